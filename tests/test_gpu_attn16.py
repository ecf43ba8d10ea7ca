"""GPU: the 16-bit matrix-pipe attention kernels (csrc/attn16.hip) and the 16-bit compute mode of the Nystrom block.

Tolerances per dtype (stated, not calibrated): operands are rounded to bf16 (8 significand bits, relative rounding 2^-9 = 2e-3
per element) or fp16 (11 bits, 2^-12 = 2.4e-4), products accumulate in fp32.  A softmax-weighted average of K such roundings
lands at BF16_TOL = 1.5e-2 / FP16_TOL = 2e-3 of the tensor's scale for outputs and gradients (max-norm); l2 errors are ~4x
smaller.  The fp32 path keeps the 1e-4 gate of tests/helpers.py."""
import pytest
import torch

import helpers
from helpers import assert_close, l2_err, params_for, rel_err, smml, synth
from oracle.nystrom import nystrom_attention
from test_gpu_parity import _load

pytestmark = pytest.mark.gpu
Fh = smml.functional
TOL = {False: 1.5e-2, True: 2e-3}      # [fp16]
GRAD_TOL_FP16 = 5e-3                   # whole-block gradients in fp16 mode (bf16 gradient products, see test_nystrom_16bit_mode_vs_oracle)


def _ref_attention(q, k, v, scale, residual=None, merged=False):
    s = (q @ k.transpose(-1, -2)) * scale
    o = torch.softmax(s, dim=-1) @ v
    if merged:
        B, H, L, D = o.shape
        o = o.permute(0, 2, 1, 3).reshape(B, L, H * D)
        if residual is not None:
            o = o + residual
    return o


@pytest.mark.parametrize("fp16", [False, True])
@pytest.mark.parametrize("B,H,Lq,Lk", [(1, 2, 1, 1), (2, 3, 33, 70), (1, 8, 300, 256), (2, 8, 256, 1000), (1, 2, 129, 31), (1, 1, 2000, 16)])
def test_attention16_vs_fp64(cuda, fp16, B, H, Lq, Lk):
    """Ragged query / key counts (partial tiles, a single key, more query slices than tiles), head-major output."""
    gen = torch.Generator().manual_seed(Lq * 1000 + Lk)
    q = torch.randn(B, H, Lq, 64, generator=gen); k = torch.randn(B, H, Lk, 64, generator=gen); v = torch.randn(B, H, Lk, 64, generator=gen)
    wo = torch.randn(B, H, Lq, 64, generator=gen)
    ref = [t.clone().double().requires_grad_() for t in (q, k, v)]
    o64 = _ref_attention(*ref, 0.125)
    (o64 * wo.double()).sum().backward()
    dev = [t.clone().to(cuda).requires_grad_() for t in (q, k, v)]
    o = Fh.attention16(*dev, scale=0.125, fp16=fp16)
    (o * wo.to(cuda)).sum().backward()
    tol = TOL[fp16]
    tag = f"attn16[{'f16' if fp16 else 'bf16'}] {B}x{H}x{Lq}x{Lk}"
    assert_close(tag + " out", o, o64, tol)
    for n, a, b in zip("qkv", dev, ref):
        if float(b.grad.abs().max()) < 1e-12:                     # one key: d q = scale P (dP - delta) k is exactly 0; what is left is
            assert float(a.grad.abs().max()) < 4 * tol            # the operand rounding of dP = V dO^T against delta = dO . O (|dP| ~ 8)
            continue
        assert_close(tag + " d" + n, a.grad, b.grad, tol)
        assert l2_err(a.grad, b.grad) <= tol / 2


@pytest.mark.parametrize("fp16", [False, True])
@pytest.mark.parametrize("B,H,Lq,Lk,merged", [(1, 2, 1, 1, False), (2, 3, 300, 70, False), (1, 8, 1000, 256, True), (2, 4, 257, 33, True),
                                             (1, 2, 513, 255, False)])
def test_attention16_fewkeys_forward(cuda, fp16, B, H, Lq, Lk, merged):
    """The two-pass forward for <= 256 keys (all keys resident in LDS, exact row maximum, no online rescale - the [n', m] side of the
    Nystrom block), forced on whatever the grid size: against fp64 AND against the online-softmax kernel on the same inputs (the
    two must agree to the operand rounding; the saved log-sum-exp too, since the shared backward consumes it); ragged last key tile,
    a single key, more than one 256-query workgroup, heads-merged output accumulated onto a residual."""
    gen = torch.Generator().manual_seed(Lq * 7 + Lk)
    q = torch.randn(B, H, Lq, 64, generator=gen); k = torch.randn(B, H, Lk, 64, generator=gen); v = torch.randn(B, H, Lk, 64, generator=gen)
    res = torch.randn(B, Lq, H * 64, generator=gen) if merged else None
    o64 = _ref_attention(q.double(), k.double(), v.double(), 0.125, res.double() if merged else None, merged=merged)
    L = smml.lib()
    outs = {}
    for mode in (0, 2):
        L.smml_attn16_set_fewkeys(mode)
        try:
            dev = [t.clone().to(cuda).requires_grad_() for t in (q, k, v)]
            o = Fh.attention16(*dev, scale=0.125, fp16=fp16, merged=merged, residual=res.to(cuda) if merged else None)
            o.sum().backward()
            outs[mode] = (o.detach(), [t.grad for t in dev])
        finally:
            L.smml_attn16_set_fewkeys(-1)
    tol = TOL[fp16]
    assert_close(f"fewkeys out vs fp64 {Lq}x{Lk}", outs[2][0], o64, tol)
    assert_close(f"fewkeys out vs online kernel {Lq}x{Lk}", outs[2][0], outs[0][0], tol / 4)
    for a, b, n in zip(outs[2][1], outs[0][1], "qkv"):           # same backward kernels, fed with each forward's output / lse
        # the two forwards hand the backward outputs / log-sum-exps that differ by their P roundings (online vs exact maximum); the
        # gradients then agree to half the fp64 gate (measured 0.9 of tol / 4 on the 1000 x 256 case)
        assert float((a - b).abs().max()) <= (tol / 2) * max(float(b.abs().max()), 1e-3), f"d{n} differs between the two forwards"


def test_attention16_merged_layout_and_residual(cuda):
    gen = torch.Generator().manual_seed(5)
    B, H, Lq, Lk = 2, 8, 200, 96
    q = torch.randn(B, H, Lq, 64, generator=gen); k = torch.randn(B, H, Lk, 64, generator=gen); v = torch.randn(B, H, Lk, 64, generator=gen)
    res = torch.randn(B, Lq, H * 64, generator=gen); wo = torch.randn(B, Lq, H * 64, generator=gen)
    ref = [t.clone().double().requires_grad_() for t in (q, k, v, res)]
    o64 = _ref_attention(ref[0], ref[1], ref[2], 0.125, ref[3], merged=True)
    (o64 * wo.double()).sum().backward()
    dev = [t.clone().to(cuda).requires_grad_() for t in (q, k, v, res)]
    o = Fh.attention16(dev[0], dev[1], dev[2], scale=0.125, fp16=True, merged=True, residual=dev[3])
    (o * wo.to(cuda)).sum().backward()
    assert_close("attn16 merged out", o, o64, TOL[True])
    for n, a, b in zip(("q", "k", "v"), dev, ref):
        assert_close("attn16 merged d" + n, a.grad, b.grad, TOL[True])
    assert torch.equal(dev[3].grad.cpu(), wo)                     # the residual's gradient is the incoming one, untouched
    assert torch.equal(res, dev[3].detach().cpu())               # and the residual itself is not modified in place


@pytest.mark.parametrize("mode,B,n", [("bf16", 2, 1000), ("fp16", 1, 4096), ("bf16", 1, 300)])
def test_nystrom_16bit_mode_vs_oracle(cuda, mode, B, n):
    """NystromAttention(compute_dtype=...) against the fp64 oracle, tolerance per dtype; the same module in its default
    fp32 mode stays within 1e-4 on the same inputs (the two modes share every parameter)."""
    tag = f"nys16:{mode}:{B}:{n}"
    mod = smml.NystromAttention(dim=512, dim_head=64, heads=8, num_landmarks=256, dropout=0.1, compute_dtype=mode)
    params = params_for(mod, 41, tag)
    mod = _load(mod, params, cuda)
    x = synth.normal((B, n, 512), 41, tag + ":x") * 0.5
    wo = synth.normal((B, n, 512), 41, tag + ":wo")
    pr = {k: v.clone().double().requires_grad_() for k, v in params.items()}
    xr = x.clone().double().requires_grad_()
    o64 = nystrom_attention(xr, pr, heads=8, dim_head=64, num_landmarks=256)
    (o64 * wo.double()).sum().backward()
    xd = x.to(cuda).requires_grad_()
    out = mod(xd)
    (out * wo.to(cuda)).sum().backward()
    tol = TOL[mode == "fp16"]
    assert mod.matrix_pipe(xd.dtype) == ("f16" if mode == "fp16" else "bf16")
    assert_close(tag + " out", out, o64, tol)
    # gradients: bf16 mode tol; fp16 mode GRAD_TOL_FP16 - its gradient products (dx = dy W, dW = dy^T x of the two projections, and the
    # landmark-side chain) run on bf16 operands since round 4 (fp16 has no range for gradients without a loss scale; the forward products
    # keep fp16's 11 bits): measured dx 3.1e-3 at 1 x 4096, 3.4e-3 at 1 x 50 000 (round 3, exact fp32 projections: 1.6e-3)
    gtol = tol if mode != "fp16" else GRAD_TOL_FP16
    assert_close(tag + " dx", xd.grad, xr.grad, gtol)
    for k, p in mod.named_parameters():
        assert_close(tag + " d" + k, p.grad, pr[k].grad, 2 * gtol)      # weight gradients sum B n' rounded products
    mod32 = _load(smml.NystromAttention(dim=512, dim_head=64, heads=8, num_landmarks=256, dropout=0.1), params, cuda)
    with torch.no_grad():
        assert_close(tag + " fp32 mode out", mod32(x.to(cuda)), o64, 1e-4)


def test_nystrom_16bit_mode_follows_the_bag_dtype(cuda):
    """A bf16 / fp16 bag selects the 16-bit compute mode (BASELINE configs 2, 5); compute_dtype='fp32' keeps the exact path."""
    torch.manual_seed(0)
    mod = smml.NystromAttention(dim=512, dim_head=64, heads=8, num_landmarks=256).to(cuda).eval()
    x = torch.randn(1, 700, 512, device=cuda) * 0.5
    with torch.no_grad():
        ref = mod(x)
        o16 = mod(x.to(torch.bfloat16))
        assert o16.dtype == torch.float32 and mod.matrix_pipe(torch.bfloat16) == "bf16" and mod.matrix_pipe(x.dtype) == "f32"
        assert rel_err(o16, ref) < 3e-2 and not torch.equal(o16, ref)
        mod.compute_dtype = "fp32"
        assert torch.equal(mod(x.to(torch.bfloat16)), mod(x.to(torch.bfloat16).float()))


@pytest.mark.parametrize("cfg", [dict(residual=False), dict(num_landmarks=64), dict(heads=4, dim=256), dict(x_fp32=True), dict(B=3, n=513),
                                 dict(num_landmarks=96, heads=2, dim=128, n=500), dict(B=1, n=40, num_landmarks=48, heads=1, dim=64)])
def test_nystrom_bf16_storage_configurations(cuda, cfg):
    """The bf16-storage pipeline (functional.py: qkv_project16 -> attention16_keys_long -> resconv16 -> attention16_queries_long) in the
    configurations the main comparison does not visit: no residual convolution (attention16_keys_long then WRITES dv), 64 landmarks (the
    pseudo-inverse chain through the generic GEMM, 64 keys on the queries-long side), 4 heads, an fp32 bag with compute_dtype='bf16' (the
    input gradient comes back as fp32), three bags of a length that needs padding, landmark counts that leave ragged key tiles (96, 48), one head, a bag shorter than the landmark count - each against the fp64 oracle, and each checked to have
    taken the bf16-storage branch."""
    import importlib
    cfg = dict(cfg)
    B, n = cfg.pop("B", 2), cfg.pop("n", 700)
    x_fp32 = cfg.pop("x_fp32", False)
    dim, heads = cfg.pop("dim", 512), cfg.pop("heads", 8)
    m = cfg.pop("num_landmarks", 256)
    residual = cfg.pop("residual", True)
    tag = f"nys16cfg:{B}:{n}:{dim}:{heads}:{m}:{int(residual)}:{int(x_fp32)}"
    mod = smml.NystromAttention(dim=dim, dim_head=64, heads=heads, num_landmarks=m, residual=residual, dropout=0.1, compute_dtype="bf16")
    params = params_for(mod, 43, tag)
    mod = _load(mod, params, cuda)
    x = synth.normal((B, n, dim), 43, tag + ":x") * 0.5
    wo = synth.normal((B, n, dim), 43, tag + ":wo")
    pr = {k: v.clone().double().requires_grad_() for k, v in params.items()}
    xr = x.clone().double().requires_grad_()
    o64 = nystrom_attention(xr, pr, heads=heads, dim_head=64, num_landmarks=m, residual=residual)
    (o64 * wo.double()).sum().backward()
    calls = []
    orig = Fh.qkv_project16
    Fh.qkv_project16 = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        xd = (x if x_fp32 else x.to(torch.bfloat16)).to(cuda).requires_grad_()
        out = mod(xd)
        (out * wo.to(cuda)).sum().backward()
    finally:
        Fh.qkv_project16 = orig
    assert calls, "the bf16-storage branch was not taken"
    assert out.dtype == torch.float32 and xd.grad.dtype == xd.dtype
    tol = TOL[False]                        # (a bf16 bag is rounded once more than the oracle's: inside the same gate)
    assert_close(tag + " out", out, o64, tol)
    assert_close(tag + " dx", xd.grad, xr.grad, tol)
    for k, p in mod.named_parameters():
        assert_close(tag + " d" + k, p.grad, pr[k].grad, 2 * tol)


@pytest.mark.parametrize("fp16", [False, True])
def test_attention16_large_score_range(cuda, fp16):
    """Scores spanning hundreds of log2 units with the row maximum in a LATE key tile: the lazy softmax reference (moved only when a tile
    maximum exceeds it by 2^8) must rescale then, stay finite, and agree with fp64 - forward and backward - incl. rows whose maximum grows
    tile after tile."""
    gen = torch.Generator().manual_seed(9)
    B, H, Lq, Lk = 1, 2, 96, 200
    q = torch.randn(B, H, Lq, 64, generator=gen) * 3.0
    k = torch.randn(B, H, Lk, 64, generator=gen) * 3.0
    k[:, :, :, :] *= torch.linspace(0.2, 2.5, Lk).view(1, 1, Lk, 1)          # later keys carry larger scores
    v = torch.randn(B, H, Lk, 64, generator=gen)
    wo = torch.randn(B, H, Lq, 64, generator=gen)
    ref = [t.clone().double().requires_grad_() for t in (q, k, v)]
    o64 = _ref_attention(*ref, 0.125)
    (o64 * wo.double()).sum().backward()
    dev = [t.clone().to(cuda).requires_grad_() for t in (q, k, v)]
    o = Fh.attention16(*dev, scale=0.125, fp16=fp16)
    (o * wo.to(cuda)).sum().backward()
    assert torch.isfinite(o).all() and all(torch.isfinite(t.grad).all() for t in dev)
    # operand rounding moves a near-one-hot row's winner only rarely; the gate is the mode's, on the tensor's scale
    tol = 4 * TOL[fp16]
    assert_close(f"large range out fp16={fp16}", o, o64, tol)
    for t, r, n in zip(dev, ref, "qkv"):
        assert_close(f"large range d{n} fp16={fp16}", t.grad, r.grad, tol)


@pytest.mark.parametrize("b,n", [(1, 33), (2, 700), (1, 257), (3, 512)])
def test_queries_long_two_query_blocks_per_wave(cuda, b, n):
    """attn16_fwd_q2_kernel (two 32-query blocks per wave) against the one-block kernel on ragged lengths: each block
    runs the same MFMA / softmax sequence on the same fragments, so outputs and log-sum-exps are the same BITS - with and without the residual."""
    from importlib import import_module
    capi = import_module("subspace-multimodal-learning_amd._capi")
    L = capi.lib()
    h, d, m = 8, 64, 256
    gen = torch.Generator().manual_seed(3 + n)
    qkv = (torch.randn(b, n, 3 * h * d, generator=gen) * 0.7).to(torch.bfloat16).to(cuda)       # token-major [b, n, (3 h d)]
    kl = (torch.randn(b, h, m, d, generator=gen) * 0.7).to(cuda)
    w = torch.randn(b, h, m, d, generator=gen).to(cuda)
    res = torch.randn(b, n, h * d, generator=gen).to(torch.bfloat16).to(cuda)
    wo = torch.randn(b, n, h * d, generator=gen).to(cuda)
    outs = {}
    try:
        for mode in (0, 2):
            L.smml_attn16_set_query_blocks(mode)
            q = qkv.clone().requires_grad_()
            o_res = Fh.attention16_queries_long(q, kl, w, res, heads=h, scale=0.125)
            (o_res.float() * wo).sum().backward()          # the backward reads the forward's log-sum-exp: equal gradients = equal lse
            outs[mode] = (o_res.detach(), Fh.attention16_queries_long(qkv, kl, w, None, heads=h, scale=0.125), q.grad[:, :, :h * d].clone())
    finally:
        L.smml_attn16_set_query_blocks(1)
    for a, bb in zip(outs[0], outs[2]):
        assert torch.equal(a, bb)
    assert torch.isfinite(outs[2][0].float()).all()
