"""GPU: train-step glue - the gradient-modulation kernel against the reference's own block
(golden) and the oracle, and the pinned double-buffered bag stager."""
import os

import numpy as np
import pytest
import torch

from helpers import assert_close, smml
from oracle.trainstep import gradient_modulate as oracle_modulate
from test_oracle_golden import gradmod_case

pytestmark = pytest.mark.gpu


def _classifier(W, b, G, dev):
    cls = torch.nn.Linear(W.shape[1], W.shape[0]).to(dev)
    cls.weight.data.copy_(W); cls.bias.data.copy_(b); cls.weight.grad = G.clone().to(dev)
    return cls


def test_gradient_modulate_golden(cuda):
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gradmod_b8.npz"))
    for seed in range(1, 9):
        ft, fi, W, b, label, G = gradmod_case(seed)
        cls = _classifier(W, b, G, cuda)
        info = smml.gradient_modulate(cls, ft.to(cuda), fi.to(cuda), label.to(cuda), return_info=True).cpu()
        ref = torch.from_numpy(z[f"case{seed}/grad"])
        assert_close(f"gradmod case {seed}", cls.weight.grad, ref, 1e-5)
        assert abs(float(info[2]) - float(z[f"case{seed}/ratio_t"])) <= 1e-5 * float(info[2])
        assert [bool(x) for x in info[5::2].tolist()] == list(z[f"case{seed}/changed_rows"])
        # rows the reference left alone are untouched bit for bit
        keep = ~torch.from_numpy(z[f"case{seed}/changed_rows"])
        assert torch.equal(cls.weight.grad.cpu()[keep], G[keep])


@pytest.mark.parametrize("B,C,hs", [(1, 3, 128), (5, 4, 64), (64, 4, 128), (8, 3, 200)])
def test_gradient_modulate_shapes_vs_oracle(cuda, B, C, hs):
    """Other batch sizes / class counts (grade and subtype have 3 classes) / widths, incl. a zero gradient row (cosine NaN ->
    no edit, as in the reference)."""
    gen = torch.Generator().manual_seed(B * 100 + C)
    ft, fi = torch.randn(B, hs, generator=gen), torch.randn(B, hs, generator=gen)
    W = torch.randn(C, 2 * hs, generator=gen) * 0.2; b = torch.randn(C, generator=gen) * 0.1
    G = torch.randn(C, 2 * hs, generator=gen) * 0.05
    G[C - 1, :hs] = 0.0
    label = torch.randint(0, C, (B,), generator=gen)
    ref, info = oracle_modulate(ft, fi, W, b, label, G)
    cls = _classifier(W, b, G, cuda)
    got = smml.gradient_modulate(cls, ft.to(cuda), fi.to(cuda), label.to(cuda), return_info=True).cpu()
    assert_close(f"gradmod {B}x{C}x{hs}", cls.weight.grad, ref, 1e-5)
    assert [int(x) for x in got[5::2].tolist()] == info["branch"]
    assert torch.equal(cls.weight.grad[C - 1].cpu(), G[C - 1])


def test_train_step_has_no_host_sync_between_backward_and_step(cuda):
    """backward -> gradient_modulate -> optimizer.step() with the CUDA sync debug mode set to 'error': any .item() /
    blocking copy in that window raises."""
    torch.manual_seed(0)
    B, hs, C = 8, 128, 4
    cls = torch.nn.Linear(2 * hs, C).to(cuda)
    opt = torch.optim.Adam(cls.parameters(), lr=1e-3, foreach=True)
    ft = torch.randn(B, hs, device=cuda, requires_grad=True); fi = torch.randn(B, hs, device=cuda, requires_grad=True)
    label = torch.randint(0, C, (B,), device=cuda)
    loss = torch.nn.functional.cross_entropy(cls(torch.cat((ft, fi), 1)), label)
    opt.zero_grad()
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        loss.backward()
        smml.gradient_modulate(cls, ft, fi, label)
        opt.step()
    finally:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    assert torch.isfinite(cls.weight).all()


def test_pinned_bag_stager(cuda):
    """Batches arrive on the device with the host values, in order, through two pinned buffers; the bf16 option narrows the
    bag on the host; buffers are reused (no allocation per step after the second batch)."""
    torch.manual_seed(0)
    batches = [(torch.randn(2, 300, 64), torch.randn(2, 7), torch.tensor([i, i + 1])) for i in range(5)]
    st = smml.PinnedBagStager(batches, cuda)
    ptrs = set()
    for k, (bag, om, lab) in enumerate(st):
        assert bag.is_cuda and torch.equal(bag.cpu(), batches[k][0]) and torch.equal(om.cpu(), batches[k][1])
        assert torch.equal(lab.cpu(), batches[k][2])
        ptrs.add(bag.data_ptr())
        y = (bag * 2).sum()          # consumer work on the current stream
    assert k == 4 and len(ptrs) == 3 and torch.isfinite(y)       # three staging slots, reused
    # a consumer may keep the previous batch across one iteration (with two slots it was overwritten under it) - checked
    # with long-running consumer kernels in flight so that a missing device-side ordering would show
    big = [(torch.full((4, 2000, 512), float(i)), torch.tensor([i])) for i in range(6)]
    st = smml.PinnedBagStager(big, cuda)
    prev = None
    acc = torch.zeros((), device=cuda)
    w = torch.randn(2048, 2048, device=cuda)
    for k, (bag, lab) in enumerate(st):
        for _ in range(20):
            w = torch.tanh(w @ w * 1e-3)                # keeps the consumer's stream busy while the next batches are staged
        if prev is not None:
            acc = acc + (prev[0] != float(k - 1)).sum() + (bag != float(k)).sum()      # the previous batch is still intact
        prev = (bag, lab)
    assert k == 5 and float(acc) == 0.0 and st.host_wait_s < 0.5
    st16 = smml.PinnedBagStager(batches[:2], cuda, bag_dtype=torch.bfloat16)
    for k, (bag, om, lab) in enumerate(st16):
        assert bag.dtype == torch.bfloat16 and torch.equal(bag.cpu(), batches[k][0].to(torch.bfloat16))
    assert list(smml.PinnedBagStager([], cuda)) == []


def _surv_case(seed, B=8, C=4, hs=128, all_censored=False, tied_times=True):
    gen = torch.Generator().manual_seed(seed)
    ft, fi = torch.randn(B, hs, generator=gen), torch.randn(B, hs, generator=gen)
    W = torch.randn(C, 2 * hs, generator=gen) * 0.2; b = torch.randn(C, generator=gen) * 0.1
    G = torch.randn(C, 2 * hs, generator=gen) * 0.05
    for r in range(0, C, 2):                                   # make some rows' halves point against each other (sim < 0)
        G[r, hs:] = -G[r, :hs] * 0.7 + 0.05 * torch.randn(hs, generator=gen)
    censor = (torch.rand(B, generator=gen) < 0.4).float()
    if all_censored:
        censor[:] = 1.0
    elif float(censor.mean()) == 1.0:
        censor[0] = 0.0
    time = torch.randint(1, 12 if tied_times else 10 ** 6, (B,), generator=gen).float()      # small range: ties in time occur
    return ft, fi, W, b, censor, time, G


@pytest.mark.parametrize("seed,B", [(1, 8), (2, 8), (3, 16), (4, 64), (5, 5), (6, 2)])
def test_gradient_modulate_survival_vs_oracle(cuda, seed, B):
    """task_type 'survival' (train_test.py:99-102,121-149; VERDICT r03 'missing' 3): risk = -sum_t S_t, C-index per branch by scikit-survival's
    pair rule, ratios, row edits - against the oracle's restatement (PARITY UNPINNED: scikit-survival is absent from this image), incl. ties in
    time, censored samples and batches without a comparable pair."""
    from oracle.trainstep import gradient_modulate_survival as oracle_surv
    ft, fi, W, b, censor, time, G = _surv_case(seed, B=B)
    ref, info = oracle_surv(ft, fi, W, b, censor, time, G)
    cls = _classifier(W, b, G, cuda)
    got = smml.gradient_modulate_survival(cls, ft.to(cuda), fi.to(cuda), censor.to(cuda), time.to(cuda), return_info=True).cpu()
    assert_close(f"gradmod survival seed {seed} B {B}", cls.weight.grad, ref, 1e-5)
    assert [int(x) for x in got[5::2].tolist()] == info["branch"]
    if info["cindex_t"] is not None:
        assert abs(float(got[0]) - info["cindex_t"]) <= 1e-6 and abs(float(got[1]) - info["cindex_i"]) <= 1e-6
    else:
        assert torch.equal(cls.weight.grad.cpu(), G)


def test_gradient_modulate_survival_all_censored_is_a_no_op(cuda):
    ft, fi, W, b, censor, time, G = _surv_case(7, all_censored=True)
    cls = _classifier(W, b, G, cuda)
    smml.gradient_modulate_survival(cls, ft.to(cuda), fi.to(cuda), censor.to(cuda), time.to(cuda))
    assert torch.equal(cls.weight.grad.cpu(), G)               # train_test.py:127-133: "All samples are censored" -> ratios None -> no edit


def _cindex_case(cuda, event, time, risk_t, risk_i):
    """A cohort whose RISK ORDER is known by construction: the branch vector of sample b is x_b on its first component, the classifier reads
    that component with weight 1 for every interval and no bias, so hazards = sigmoid(x_b) in every interval and
    risk = -sum_t prod (1 - hazard) grows strictly with x_b; equal x give exactly equal risks."""
    B, C, hs = len(event), 4, 8
    ft, fi = torch.zeros(B, hs), torch.zeros(B, hs)
    ft[:, 0] = torch.tensor(risk_t); fi[:, 0] = torch.tensor(risk_i)
    W = torch.zeros(C, 2 * hs); W[:, 0] = 1.0; W[:, hs] = 1.0
    b = torch.zeros(C)
    G = torch.randn(C, 2 * hs, generator=torch.Generator().manual_seed(1)) * 0.05
    cls = _classifier(W, b, G, cuda)
    censor = torch.tensor([0.0 if e else 1.0 for e in event])
    info = smml.gradient_modulate_survival(cls, ft.to(cuda), fi.to(cuda), censor.to(cuda), torch.tensor(time, dtype=torch.float32).to(cuda),
                                           return_info=True).cpu()
    return float(info[0]), float(info[1]), cls.weight.grad.cpu(), G


def test_gradient_modulate_survival_hand_computed_cindex(cuda):
    """the survival branch is parity-UNPINNED (scikit-survival is absent), so its pair rule is pinned by hand-computed cases on
    cohorts with a known risk order: ties in time, ties in risk, event / event ties, a censored sample at an event's time, all but one censored,
    and the two cases in which nothing may be modulated (flagged in info: -2 all censored, -1 no comparable pair)."""
    T, F = True, False
    # perfectly ordered (tumor) / reversed (immune): 1.0 and 0.0
    ct, ci, _, _ = _cindex_case(cuda, [T, T, T], [1, 2, 3], [3.0, 2.0, 1.0], [1.0, 2.0, 3.0])
    assert ct == 1.0 and ci == 0.0
    # a tie in risk between the two earliest events counts one half: pairs (0,1) tie, (0,2) and (1,2) concordant -> 2.5 / 3
    ct, ci, _, _ = _cindex_case(cuda, [T, T, F], [1, 2, 3], [2.0, 2.0, 1.0], [3.0, 2.0, 1.0])
    assert abs(ct - 2.5 / 3) < 1e-6 and ci == 1.0
    # event and censoring at the same time: comparable (1 pair); two events at the same time: not comparable with each other
    ct, ci, _, _ = _cindex_case(cuda, [T, F], [5, 5], [1.0, 0.0], [0.0, 1.0])
    assert ct == 1.0 and ci == 0.0
    ct, ci, _, _ = _cindex_case(cuda, [T, T, F], [5, 5, 9], [2.0, 1.0, 0.0], [2.0, 1.0, 3.0])
    assert ct == 1.0 and ci == 0.0                    # 2 comparable pairs each: (0,2), (1,2)
    # all but one censored: only the event sample opens pairs, with everyone who outlives it
    ct, ci, _, _ = _cindex_case(cuda, [F, T, F, F], [1, 2, 3, 4], [9.0, 2.0, 1.0, 3.0], [0.0, 2.0, 2.0, 1.0])
    assert abs(ct - 0.5) < 1e-6 and abs(ci - 0.75) < 1e-6          # pairs (1,2), (1,3): tumor 1 concordant 1 discordant; immune 1 tie 1 concordant
    # nothing to modulate: every sample censored (-2), no comparable pair (-1: the only event is the last to leave) - gradients untouched
    ct, ci, g, G = _cindex_case(cuda, [F, F, F], [1, 2, 3], [1.0, 2.0, 3.0], [1.0, 2.0, 3.0])
    assert ct == -2.0 and ci == -2.0 and torch.equal(g, G)
    ct, ci, g, G = _cindex_case(cuda, [F, F, T], [1, 2, 3], [1.0, 2.0, 3.0], [1.0, 2.0, 3.0])
    assert ct == -1.0 and ci == -1.0 and torch.equal(g, G)
