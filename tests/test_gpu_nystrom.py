"""GPU parity of the Nystrom block (a1-a3): HIP kernels vs golden vectors generated from the reference and vs
the CPU oracle (fp64-calibrated tolerance, see test_gpu_parity._calibrated)."""
import argparse

import pytest
import torch

from helpers import Golden, assert_close, params_for, rel_err, smml, synth
from oracle.nystrom import nystrom_attention, pinv_newton_schulz, ppeg, trans_layer, trans_mil
from test_gpu_parity import _assert_close, _calibrated, _load

pytestmark = pytest.mark.gpu
Fh = smml.functional


def test_matmul4_autograd_all_layouts(cuda):
    torch.manual_seed(0)
    for ta in (False, True):
        for tb in (False, True):
            for bcast in (False, True):
                A = torch.randn(2, 1 if bcast else 3, *((13, 70) if ta else (70, 13)))
                B = torch.randn(1 if bcast else 2, 3, *((40, 13) if tb else (13, 40)))
                R = torch.randn(2, 3, 70, 40)
                W = torch.randn(2, 3, 70, 40)
                def f(a, b, r, mm):
                    return mm(a, b, r)
                ref_in = [t.clone().requires_grad_() for t in (A, B, R)]
                oa = ref_in[0].transpose(-1, -2) if ta else ref_in[0]
                ob = ref_in[1].transpose(-1, -2) if tb else ref_in[1]
                ref = -0.5 * (oa @ ob) + 3.0 * ref_in[2]
                (ref * W).sum().backward()
                dev_in = [t.clone().to(cuda).requires_grad_() for t in (A, B, R)]
                got = Fh.matmul4(dev_in[0], dev_in[1], dev_in[2], ta=ta, tb=tb, alpha=-0.5, beta=3.0)
                (got * W.to(cuda)).sum().backward()
                _assert_close(f"C ta={ta} tb={tb} bc={bcast}", got, ref, 1e-5)
                for n, g, r in zip("ABR", dev_in, ref_in):
                    _assert_close(f"d{n} ta={ta} tb={tb} bc={bcast}", g.grad, r.grad, 1e-5)
    # merged-heads output
    A = torch.randn(2, 4, 50, 16); B = torch.randn(2, 4, 16, 8); R = torch.randn(2, 50, 32)
    ref = (A @ B).permute(0, 2, 1, 3).reshape(2, 50, 32) + R
    got = Fh.matmul4(A.to(cuda), B.to(cuda), R.to(cuda), merged=True)
    _assert_close("merged", got, ref, 1e-5)


def test_softmax_segment_mean_resconv(cuda):
    torch.manual_seed(1)
    for L in (16, 300, 5000):
        x = torch.randn(7, L) * 3
        w = torch.randn(7, L)
        xr = x.clone().requires_grad_(); (torch.softmax(xr, -1) * w).sum().backward()
        xd = x.clone().to(cuda).requires_grad_(); y = Fh.softmax_rows(xd); (y * w.to(cuda)).sum().backward()
        _assert_close(f"softmax L={L}", y, torch.softmax(x, -1), 1e-5); _assert_close(f"dsoftmax L={L}", xd.grad, xr.grad, 1e-5)
    x = torch.randn(2, 3, 40, 8); w = torch.randn(2, 3, 8, 8)
    xr = x.clone().requires_grad_(); (xr.reshape(2, 3, 8, 5, 8).mean(3) * w).sum().backward()
    xd = x.clone().to(cuda).requires_grad_(); y = Fh.segment_mean(xd, 5); (y * w.to(cuda)).sum().backward()
    _assert_close("segment_mean", y, x.reshape(2, 3, 8, 5, 8).mean(3), 1e-5); _assert_close("dsegment_mean", xd.grad, xr.grad, 1e-5)
    v = torch.randn(2, 4, 50, 8); k = torch.randn(4, 1, 33, 1); wo = torch.randn(2, 50, 32)
    vr, kr = v.clone().requires_grad_(), k.clone().requires_grad_()
    ref = torch.nn.functional.conv2d(vr, kr, padding=(16, 0), groups=4).permute(0, 2, 1, 3).reshape(2, 50, 32)
    (ref * wo).sum().backward()
    vd, kd = v.clone().to(cuda).requires_grad_(), k.clone().to(cuda).requires_grad_()
    got = Fh.resconv(vd, kd); (got * wo.to(cuda)).sum().backward()
    _assert_close("resconv", got, ref, 1e-5); _assert_close("dv", vd.grad, vr.grad, 1e-5); _assert_close("dw", kd.grad, kr.grad, 1e-5)


@pytest.mark.parametrize("tag,B,n,dim,dh,m", [("nystrom_n37_m16", 2, 37, 64, 8, 16), ("nystrom_n64_m16", 2, 64, 64, 8, 16),
                                               ("nystrom_n257_m256", 1, 257, 512, 64, 256)])
def test_nystrom_golden(cuda, tag, B, n, dim, dh, m):
    g = Golden(tag)
    mod = smml.NystromAttention(dim=dim, dim_head=dh, heads=8, num_landmarks=m, pinv_iterations=6, residual=True, dropout=0.1)
    mod = _load(mod, params_for(mod, 42, tag), cuda)
    x = synth.normal((B, n, dim), 42, tag + ":x").to(cuda).requires_grad_()
    w_out = synth.normal((B, n, dim), 42, tag + ":wout").to(cuda)
    out = mod(x)
    (out * w_out).sum().backward()
    g.check("out", out); g.check("dx", x.grad)
    for k, p in mod.named_parameters():
        g.check("grad:" + k, p.grad, what="d" + k)


@pytest.mark.parametrize("tag,B,n,dim,dh,m", [("nystrom_masked_n37_m16", 2, 37, 64, 8, 16), ("nystrom_masked_n64_m16", 2, 64, 64, 8, 16)])
def test_nystrom_masked_golden(cuda, tag, B, n, dim, dh, m):
    """The `mask` argument (models/NystromAttention.py:84,92-96,106-118,127-133) against the reference's own output; a bf16 bag with a mask takes
    the exact path too."""
    from test_oracle_golden import option_masks
    g = Golden(tag)
    mod = smml.NystromAttention(dim=dim, dim_head=dh, heads=8, num_landmarks=m, pinv_iterations=6, residual=True, dropout=0.1)
    mod = _load(mod, params_for(mod, 42, tag), cuda)
    x = synth.normal((B, n, dim), 42, tag + ":x").to(cuda).requires_grad_()
    w_out = synth.normal((B, n, dim), 42, tag + ":wout").to(cuda)
    mask = ~option_masks(tag + ":mask", B, n); mask[0, :5] = False
    out = mod(x, mask=mask.to(cuda))
    (out * w_out).sum().backward()
    g.check("out", out); g.check("dx", x.grad)
    for k, p in mod.named_parameters():
        g.check("grad:" + k, p.grad, what="d" + k)


def test_pinv_translayer_ppeg_golden(cuda):
    a2 = torch.softmax(synth.normal((2, 3, 16, 16), 42, "pinv:x"), dim=-1)
    Golden("pinv_m16").check("z", smml.moore_penrose_iter_pinv(a2.to(cuda), 6))
    dim = 64
    g = Golden("translayer_d64")
    tl = smml.TransLayer(dim=dim)
    tl = _load(tl, params_for(tl, 42, "translayer"), cuda)
    x = synth.normal((2, 37, dim), 42, "translayer:x").to(cuda).requires_grad_()
    w = synth.normal((2, 37, dim), 42, "translayer:w").to(cuda)
    out = tl(x); (out * w).sum().backward()
    g.check("out", out); g.check("dx", x.grad)
    for k, p in tl.named_parameters():
        g.check("grad:" + k, p.grad, what="d" + k)
    pp = smml.PPEG(dim=dim)
    pp = _load(pp, params_for(pp, 42, "ppeg"), cuda)
    Golden("ppeg_d64").check("out", pp(x.detach(), 6, 6))
    # PPEG gradients vs the oracle
    xp = synth.normal((2, 37, dim), 5, "ppeg:x"); wp = synth.normal((2, 37, dim), 5, "ppeg:w")
    pr = {k: v.detach().cpu().clone().requires_grad_() for k, v in pp.state_dict().items()}
    xr = xp.clone().requires_grad_(); (ppeg(xr, 6, 6, pr) * wp).sum().backward()
    xd = xp.clone().to(cuda).requires_grad_(); pp.zero_grad(); (pp(xd, 6, 6) * wp.to(cuda)).sum().backward()
    _assert_close("ppeg dx", xd.grad, xr.grad, 1e-5)
    for k, p in pp.named_parameters():
        _assert_close("ppeg d" + k, p.grad, pr[k].grad, 2e-5)


def test_transmil_vs_oracle(cuda):
    """Single-modality Nystrom MIL (BASELINE config 1 shape family): 60 instances x 1024 -> 8x8 wrap-padded grid + cls."""
    args = argparse.Namespace(label_dim=4, path_dim=128, input_path_dim=1024)
    net = smml.TransMIL(args)
    params = params_for(net, 9, "transmil")
    net = _load(net, params, cuda)
    x = synth.bag(1, 60, 1024, 9, "transmil:bag")
    run = {}
    for dt in (torch.float32, torch.float64):
        pref = {k: v.clone().to(dt).requires_grad_() for k, v in params.items()}
        enc, logits = trans_mil(x.to(dt), pref)
        (enc.sum() + logits.pow(2).sum()).backward()
        run[dt] = (enc, logits, pref)
    enc, logits, _ = net(x.to(cuda))
    (enc.sum() + logits.pow(2).sum()).backward()
    r32, r64 = run[torch.float32], run[torch.float64]
    _calibrated("encoded", enc, r32[0], r64[0]); _calibrated("logits", logits, r32[1], r64[1])
    for k, p in net.named_parameters():
        if r32[2][k].grad is None:
            continue
        _calibrated("d" + k, p.grad, r32[2][k].grad, r64[2][k].grad)


def test_transmil_bf16_compute_mode_vs_oracle(cuda):
    """TransMIL with args.nystrom_compute_dtype = 'bf16' (both TransLayers in the block's bf16 compute mode, everything else exact): against
    the fp64 oracle at twice the block's own bf16 gate (two blocks in series), 500 instances -> 23 x 23 grid + cls, 530 tokens padded to 768."""
    args = argparse.Namespace(label_dim=4, path_dim=128, input_path_dim=1024, nystrom_compute_dtype="bf16")
    net = smml.TransMIL(args)
    params = params_for(net, 10, "transmil16")
    net = _load(net, params, cuda)
    x = synth.bag(2, 500, 1024, 10, "transmil16:bag")
    pref = {k: v.clone().double().requires_grad_() for k, v in params.items()}
    enc64, log64 = trans_mil(x.double(), pref)
    (enc64.sum() + log64.pow(2).sum()).backward()
    enc, logits, _ = net(x.to(cuda))
    (enc.sum() + logits.pow(2).sum()).backward()
    tol = 3e-2
    assert_close("transmil bf16 mode encoded", enc, enc64, tol)
    assert_close("transmil bf16 mode logits", logits, log64, tol)
    for k, p in net.named_parameters():
        if pref[k].grad is not None:
            assert_close("transmil bf16 mode d" + k, p.grad, pref[k].grad, 2 * tol)


def test_nystrom_long_bag_self_consistency(cuda):
    """n = 10 000 x 512, m = 256 (front padding to 10 240, l = 40): rows of a1 z a3 sum to ~1 is NOT guaranteed by
    the approximation, but linearity in v is: out(v-weights scaled) relation via to_qkv's v block; and bag independence
    does not hold (batch-global pinv max) - so check finite values, shape, and equality of two identical bags."""
    torch.manual_seed(0)
    mod = smml.NystromAttention(dim=512, dim_head=64, heads=8, num_landmarks=256).to(cuda).eval()
    x1 = torch.randn(1, 10000, 512, device=cuda) * 0.5
    x = torch.cat((x1, x1), 0).requires_grad_()
    out = mod(x)
    assert out.shape == (2, 10000, 512) and torch.isfinite(out).all()
    assert torch.equal(out[0], out[1])
    out.pow(2).mean().backward()
    assert torch.isfinite(x.grad).all() and float((x.grad[0] - x.grad[1]).abs().max()) <= 1e-6 * float(x.grad.abs().max())


@pytest.mark.parametrize("B,n,dim,dh,m", [(2, 100, 64, 8, 32), (1, 500, 128, 16, 64), (3, 33, 64, 8, 16), (1, 64, 64, 8, 64),
                                          (2, 1000, 512, 64, 256)])
def test_nystrom_vs_oracle_shapes(cuda, B, n, dim, dh, m):
    """Shapes outside the golden set: n below / equal / far above the landmark count, front padding of every size,
    the reference's dim_head 64 / 256 landmarks with a long reduction (deterministic split-K path); outputs and all
    gradients against the oracle (fp32 run, fp64-calibrated tolerance)."""
    tag = f"nys:{B}:{n}:{dim}:{m}"
    mod = smml.NystromAttention(dim=dim, dim_head=dh, heads=8, num_landmarks=m, pinv_iterations=6, residual=True, dropout=0.1)
    params = params_for(mod, 11, tag)
    mod = _load(mod, params, cuda)
    x = synth.normal((B, n, dim), 11, tag + ":x") * 0.5
    wo = synth.normal((B, n, dim), 11, tag + ":wo")
    run = {}
    for dt in (torch.float32, torch.float64):
        pr = {k: v.clone().to(dt).requires_grad_() for k, v in params.items()}
        xr = x.clone().to(dt).requires_grad_()
        o = nystrom_attention(xr, pr, heads=8, dim_head=dh, num_landmarks=m)
        (o * wo.to(dt)).sum().backward()
        run[dt] = (o, xr.grad, pr)
    xd = x.to(cuda).requires_grad_()
    out = mod(xd)
    (out * wo.to(cuda)).sum().backward()
    r32, r64 = run[torch.float32], run[torch.float64]
    _calibrated("out", out, r32[0], r64[0]); _calibrated("dx", xd.grad, r32[1], r64[1])
    for k, p in mod.named_parameters():
        _calibrated("d" + k, p.grad, r32[2][k].grad, r64[2][k].grad)


def test_cmta_golden(cuda):
    """The reference's default model (mode cmta): Transformer_P / Transformer_G encoders + decoders on the Nystrom kernels,
    P_in_G / G_in_P co-attention, concat fusion - outputs and all 104 parameter gradients against the reference's own."""
    from test_oracle_golden import CMTA_OUTPUTS, cmta_inputs, cmta_params
    g = Golden("cmta_n150")
    net = smml.CMTA(argparse.Namespace(label_dim=4))
    net = _load(net, cmta_params(net), cuda)
    x_path, x_omic, w = cmta_inputs()
    xp, xo = x_path.to(cuda).requires_grad_(), x_omic.to(cuda).requires_grad_()
    out = net(x_path=xp, x_omic=xo)
    loss = (out[0] * w[0].to(cuda)).sum() + sum((out[2 + i] * w[i].to(cuda)).sum() for i in range(1, 5))
    loss.backward()
    for nm, o in zip(CMTA_OUTPUTS, out):
        g.check(nm, o)
    g.check("dx_path", xp.grad); g.check("dx_omic", xo.grad)
    assert abs(loss.item() - g.scalar("loss")) <= 1e-4 * abs(g.scalar("loss"))
    with_grad = {k for k, p in net.named_parameters() if p.grad is not None}
    assert with_grad == {k[5:] for k in g.keys("grad:")}
    for k, p in net.named_parameters():
        g.check("grad:" + k, p.grad, what="d" + k)


def test_cmta_bf16_compute_mode_close_to_exact(cuda):
    """CMTA with args.nystrom_compute_dtype = 'bf16' (the eight Nystrom blocks of its four transformers in bf16 compute mode) against the same
    model on the exact path, same parameters and inputs: logits and hazards within 3e-2 of their scale, every parameter that gets a gradient
    on the exact path gets one in bf16 mode."""
    from test_oracle_golden import cmta_inputs, cmta_params
    x_path, x_omic, w = cmta_inputs()
    outs = {}
    for cd in (None, "bf16"):
        net = smml.CMTA(argparse.Namespace(label_dim=4, nystrom_compute_dtype=cd))
        net = _load(net, cmta_params(net), cuda)
        out = net(x_path=x_path.to(cuda), x_omic=x_omic.to(cuda))
        (out[0] * w[0].to(cuda)).sum().backward()
        outs[cd] = ([o.detach() for o in out], {k for k, p in net.named_parameters() if p.grad is not None})
    for a, b in zip(outs["bf16"][0], outs[None][0]):
        if torch.is_tensor(a) and a.is_floating_point():
            assert_close("cmta bf16 mode vs exact", a, b.double().cpu(), 3e-2)
    assert outs["bf16"][1] == outs[None][1]


def test_cmta_reference_bag_size_train_mode(cuda):
    """CMTA on the reference's bag size (8 bags of 2500 x 1024, config_mine.yaml:2,38) in train() mode (Dropout 0.25 on the
    wsi features, AlphaDropout in the SNN blocks, Nystrom output dropout 0.1): runs, finite, every parameter gets a gradient."""
    torch.manual_seed(0)
    net = smml.CMTA(argparse.Namespace(label_dim=4)).to(cuda).train()
    xp = synth.bag(8, 2500, 1024, 3, "cmta:big").to(cuda); xo = synth.normal((8, 431), 3, "cmta:bigo").to(cuda)
    out = net(x_path=xp, x_omic=xo)
    assert out[0].shape == (8, 4) and out[3].shape == (8, 256)
    l1 = torch.nn.functional.l1_loss(out[3], out[4].detach()) + torch.nn.functional.l1_loss(out[5], out[6].detach())   # train_test.py:371-373
    (torch.nn.functional.cross_entropy(out[0], torch.randint(0, 4, (8,), device=cuda)) + l1).backward()
    for k, p in net.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k


@pytest.mark.parametrize("B,H,n,D,KW", [(1, 8, 5, 64, 33), (2, 8, 17, 64, 33), (1, 3, 48, 32, 33), (2, 8, 1000, 64, 33), (1, 8, 531, 16, 33),
                                        (2, 4, 70, 64, 9), (1, 2, 40, 24, 33)])
def test_resconv_strip_kernels_vs_torch(cuda, B, H, n, D, KW):
    """The 33-tap residual convolution (NystromAttention.py:62-66,144-145) on its own: the strip kernels (16 tokens per thread,
    every row read once) at lengths below / across / far beyond a strip, head dims the weight-gradient walk supports (powers of
    two) and one it does not (24: generic kernels), and a tap count that takes the generic path - against F.conv2d in fp64."""
    gen = torch.Generator().manual_seed(B * 1000 + n)
    v = torch.randn(B, H, n, D, generator=gen)
    w = torch.randn(H, 1, KW, 1, generator=gen) * 0.2
    wo = torch.randn(B, n, H * D, generator=gen)
    vd, wd = v.to(cuda).requires_grad_(), w.to(cuda).requires_grad_()
    out = Fh.resconv(vd, wd)                                            # [B, n, H*D]
    (out * wo.to(cuda)).sum().backward()
    v64, w64 = v.double().requires_grad_(), w.double().requires_grad_()
    ref = torch.nn.functional.conv2d(v64, w64, padding=(KW // 2, 0), groups=H)        # [B, H, n, D]
    ref = ref.permute(0, 2, 1, 3).reshape(B, n, H * D)
    (ref * wo.double()).sum().backward()
    tag = f"resconv B={B} H={H} n={n} D={D} KW={KW}"
    assert_close(tag + " out", out, ref, 1e-5)
    assert_close(tag + " dv", vd.grad, v64.grad, 1e-5)
    assert_close(tag + " dw", wd.grad, w64.grad, 1e-5)


def test_pinv_side_stream_is_bit_identical(cuda):
    """NystromAttention runs its pseudo-inverse on a second stream (nystrom_attention._PinvFork): with and without the overlap the
    block's output and every gradient are the same bits (the same kernels on the same data; only the stream differs)."""
    import importlib
    na = importlib.import_module(smml.__name__ + ".nystrom_attention")
    torch.manual_seed(0)
    mod = smml.NystromAttention(dim=512, dim_head=64, heads=8, num_landmarks=64, pinv_iterations=6, residual=True).to(cuda)
    x = torch.randn(2, 300, 512, generator=torch.Generator().manual_seed(4)).to(cuda)
    res = {}
    for flag in (True, False, True):
        na.PINV_OVERLAP = flag
        try:
            xi = x.clone().requires_grad_()
            mod.zero_grad(set_to_none=True)
            out = mod(xi)
            out.square().sum().backward()
            torch.cuda.synchronize()
            got = [out.detach().clone(), xi.grad.clone()] + [p.grad.clone() for p in mod.parameters()]
        finally:
            na.PINV_OVERLAP = True
        if flag in res:
            continue
        res[flag] = got
    # atomically accumulated weight gradients (split-K GEMMs, res_conv.weight) are not run-to-run identical: compare those to 1e-6
    for i, (a, b) in enumerate(zip(res[True], res[False])):
        if i < 2:
            assert torch.equal(a, b), f"tensor {i} differs between overlapped and serial execution"
        else:
            assert_close(f"param grad {i} overlapped vs serial", a, b, 1e-6)


@pytest.mark.parametrize("m,fast", [(256, 2), (256, 1), (256, 0), (64, 2), (48, 1)])
def test_newton_schulz_chain_vs_fp64(cuda, m, fast):
    """The pseudo-inverse chain (csrc/pinv_chain.hip: one C call per direction) against the same iteration in fp64 with autograd:
    m = 256 runs as two bf16 planes per matrix on the 16-bit pipe (fast = 2 with reduced precision requested: the 16-bit compute mode), on
    the exact-fp32 panel-prefetch chain kernel (fast = 1) or, like every other m, through the generic batched GEMM (fast = 0).  z0 is an
    independent input here: as a function of x it carries max(row sums) of a row-stochastic matrix - all ties - whose gradient goes
    to whichever row the max picks (tests of the whole block cover that part against the golden vectors)."""
    import importlib
    na = importlib.import_module(smml.__name__ + ".nystrom_attention")
    gen = torch.Generator().manual_seed(m + fast)
    a2 = torch.softmax(torch.randn(2, 3, m, m, generator=gen) * 0.5 + 4.0 * torch.eye(m), dim=-1)
    z0 = (a2.transpose(-1, -2) / (a2.abs().sum(-1).max() * a2.abs().sum(-2).max())).contiguous()
    wo = torch.randn(2, 3, m, m, generator=gen)
    xr, zr = a2.double().requires_grad_(), z0.double().requires_grad_()
    eye = torch.eye(m, dtype=torch.float64)
    z = zr
    for _ in range(6):                                     # models/NystromAttention.py:28-33
        xz = xr @ z
        z = 0.25 * z @ (13 * eye - xz @ (15 * eye - xz @ (7 * eye - xz)))
    (z * wo.double()).sum().backward()
    L = smml.lib()
    L.smml_newton_schulz_set_fast(fast)
    try:
        xd, zd = a2.to(cuda).requires_grad_(), z0.to(cuda).requires_grad_()
        got = na._NewtonSchulz.apply(xd, zd, 6, fast == 2)
        (got * wo.to(cuda)).sum().backward()
        torch.cuda.synchronize()
    finally:
        L.smml_newton_schulz_set_fast(-1)                  # back to the default (SMML_CHAIN_FAST or 2)
    tol = 2e-4 if (fast == 2 and m == 256) else 1e-5          # the two-plane form: 16-bit operand mantissas (2^-17 per element)
    assert_close(f"pinv chain m={m} fast={fast} z", got, z.detach(), tol)
    assert_close(f"pinv chain m={m} fast={fast} dx", xd.grad, xr.grad, tol)
    # the converged iteration forgets its start: dz0 is ~1e-6 of dx in size, so it is held to the scale of dx, not to its own
    err = float((zd.grad.double().cpu() - zr.grad).abs().max() / xr.grad.abs().max())
    assert err <= tol, f"pinv chain m={m} fast={fast}: dz0 error {err:.2e} of the scale of dx"
