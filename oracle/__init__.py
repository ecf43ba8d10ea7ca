"""CPU oracle for the multimodal-MIL attention/fusion hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped
product path: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and there only as the
checker / reported CPU baseline, never as the thing measured or shipped.

Each module is a plain-PyTorch fp32 restatement (own formulation: token-major
layouts, explicit bilinear gather, chunked position-bias evaluation) of the
algorithm in helenypzhang/Subspace-Multimodal-Learning; every function cites
the reference ``file:line`` it follows.

Pinning: the reference ships no tests / golden vectors,
so the oracle is pinned against outputs of the reference itself, generated in
the build container by ``tests/golden/make_golden.py`` (which imports
``/root/reference`` read-only) and committed as ``tests/golden/*.npz``.
``tests/test_oracle_golden.py`` re-checks the oracle against those vectors on
every CPU test run.
"""
