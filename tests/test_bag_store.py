"""Bag data format: fixdim index rule (oracle restatement of data/dataset.py:151-175 vs the package's
vectorised form - CPU - and vs the device kernel - GPU, bit-exact), the packed bf16 store round trip, the device gather."""
import os
import warnings

import numpy as np
import pytest
import torch

from helpers import smml
from oracle.bagstore import f32_to_bf16_bits, fixdim_indices as oracle_indices

CASES = [(1, 7), (3, 10), (5, 5), (7, 2500), (2500, 2500), (2499, 2500), (1250, 2500), (2501, 2500), (3750, 2500), (5000, 2500),
         (12345, 2500), (7, 3), (10, 4), (9, 6), (100003, 2500), (2500, 10000), (31337, 10000)]


def test_fixdim_index_rule_known_answers():
    # short bag: repeated floor(10 / 3) = 3 times + the first 10 % 3 = 1 rows
    assert oracle_indices(3, 10).tolist() == [0, 1, 2, 0, 1, 2, 0, 1, 2, 0]
    # long bag: int(np.around(i * 2.5)) with round-half-even: 0, 2.5 -> 2, 5, 7.5 -> 8
    assert oracle_indices(10, 4).tolist() == [0, 2, 5, 8]
    assert oracle_indices(9, 6).tolist() == [0, 2, 3, 4, 6, 8]          # 1.5 -> 2, 4.5 -> 4, 7.5 -> 8
    for n, fx in CASES:
        a, b = oracle_indices(n, fx), smml.fixdim_indices(n, fx)
        assert a.shape == (fx,) and np.array_equal(a, b), (n, fx)
        assert b.min() >= 0 and b.max() < n


def test_bf16_conversion_and_store_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1000,)).astype(np.float32) * 3
    x[:4] = [0.0, -0.0, np.inf, np.nan]
    x[4] = np.float32(1.00390625)            # exactly half way between two bf16 values: ties to even
    bits = smml.bag_store.f32_to_bf16_bits(x)
    assert np.array_equal(bits, f32_to_bf16_bits(x))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = torch.from_numpy(x).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    ok = ~np.isnan(x)
    assert np.array_equal(bits[ok], ref[ok]) and (bits[3] & 0x7F80) == 0x7F80 and (bits[3] & 0x7F) != 0
    bags = [rng.standard_normal((n, 64)).astype(np.float32) for n in (5, 300, 1)]
    path = os.path.join(tmp_path, "bags.smml")
    with smml.BagStoreWriter(path, 64) as w:
        for i, b in enumerate(bags):
            w.add(f"case{i}", b[None] if i == 1 else b)          # [1, n, dim] like the h5 Res_feature, or [n, dim]
    st = smml.BagStore(path)
    assert len(st) == 3 and st.names == ["case0", "case1", "case2"] and st.dim == 64
    for i, b in enumerate(bags):
        got = st.raw(i)
        assert got.dtype == torch.bfloat16 and got.shape == b.shape
        assert torch.equal(got, torch.from_numpy(b).to(torch.bfloat16))
        assert st.raw(i).data_ptr() % 256 == st.raw(0).data_ptr() % 256          # 256-byte aligned rows blocks
    assert torch.equal(st.bag(1, 100), st.raw(1)[torch.from_numpy(oracle_indices(300, 100))])
    assert torch.equal(st.bag("case0", 12), st.raw(0)[torch.from_numpy(oracle_indices(5, 12))])
    ds = smml.BagStoreDataset(st, fixdim=16)
    assert len(ds) == 3 and ds[2][0].shape == (16, 64)
    with pytest.raises(RuntimeError):
        smml.bag_store.read_h5_res_feature(os.path.join(tmp_path, "missing.h5"))     # h5py is not installed here
    st.close()


@pytest.mark.gpu
def test_fixdim_device_indices_bit_exact(cuda):
    for n, fx in CASES:
        dev = smml.bag_store.fixdim_indices_device(n, fx, cuda).cpu().numpy()
        assert np.array_equal(dev, oracle_indices(n, fx)), (n, fx)


@pytest.mark.gpu
def test_fixdim_gather_device(cuda):
    rng = np.random.default_rng(1)
    for n, fx, dim in ((7, 50, 64), (3000, 2500, 1024), (2500, 2500, 1024), (811, 10000, 512)):
        raw = torch.from_numpy(rng.standard_normal((n, dim)).astype(np.float32)).to(torch.bfloat16)
        idx = torch.from_numpy(oracle_indices(n, fx))
        for dt in (torch.float32, torch.bfloat16):
            out = smml.fixdim_gather(raw.to(cuda), fx, dt)
            assert out.dtype == dt and torch.equal(out.cpu(), raw[idx].to(dt)), (n, fx, dim, dt)
