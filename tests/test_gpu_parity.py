"""GPU (MI355X) parity: the HIP kernels, called through the C-ABI, against the CPU oracle on identical
seeded inputs, and against the golden vectors generated from the reference.

Tolerances (north_star): fp32 within 1e-4 relative (to the tensor's scale); integer sampling path bit-exact.
Parameter gradients that are sums over 1e7+ pairs with cancellation get 1e-3 (see test_oracle_golden)."""
import numpy as np
import pytest
import torch

import helpers
import oracle.deform as odeform
from helpers import Golden, assert_calibrated, assert_close, assert_zero_grad, decision_tap, l2_err, params_for, rel_err, smml, synth
from oracle.deform import deform_cross_attention_1d, deform_cross_attention_2d, sample_positions
from oracle.losses import batch_loss
from oracle.mil import deform_cross_trans_mil, deform_pathomic_net
from test_oracle_golden import ZERO_GRADS, pathomic_args

pytestmark = pytest.mark.gpu
Fh = smml.functional
TOL = 1e-4


def _load(mod, params, dev):
    mod.load_state_dict(params)
    return mod.to(dev).eval()


def _assert_close(name, got, ref, tol=TOL):
    assert_close(name, got, ref, tol)


def _calibrated(name, got, ref32, ref64, floor=TOL, ref32_alt=None):
    """HIP result vs the fp64 oracle under the policy of tests/helpers.py: max-norm within max(1e-4, 2 x the fp32 oracle's own
    distance to fp64) (capped at 1e-3 unless the oracle itself is further off), l2-norm within max(1e-4, 1.5 x the fp32
    oracle's own l2 distance)."""
    assert_calibrated(name, got, ref32, ref64, floor, ref32_alt)


class cpb_probe:
    """with cpb_probe() as pr: ... oracle backward ...; pr.scale(bias_param) = sum |d bias| over all (query, key) pairs."""

    def __enter__(self):
        self.d = {}
        odeform.GRAD_PROBE = self.d
        return self

    def __exit__(self, *a):
        odeform.GRAD_PROBE = None

    def scale(self, tensor):
        return self.d.get(id(tensor), 0.0)

    def cpb_units(self):
        """ReLU units of the position-bias MLP evaluated by the oracle runs inside the block (fp32 + fp64 runs: twice the
        problem's count)."""
        return self.d.get("cpb_units", 0)

    def boundary_margin(self):
        """Smallest distance of a sample position's pixel coordinate to an integer (any oracle run inside the block)."""
        return self.d.get("boundary", 1.0)


def _compare_param_grads(mod, p32, p64, skip=(), probe=None, p32_alt=None):
    """Every parameter gradient against the oracle under the plain rules of tests/helpers.py - max(1e-4, 2 x noise) in the max
    norm, max(1e-4, 1.5 x the fp32 oracle's own l2 distance to fp64) in the l2 norm - with no input-dependent exemption: the
    callers impose the kernels' own piecewise-linear decisions on both oracle runs (helpers.decision_tap), so a rounding-level
    tie of a ReLU pre-activation or of a sample position on a pixel boundary no longer moves any gradient.  `p32_alt`: the oracle's
    parameters after a second fp32 run on another back end (helpers.assert_calibrated).  All tensors are
    compared and recorded before the first failure is raised.  `rel_pos_bias.mlp.2.bias` is exactly zero in exact arithmetic
    (softmax shift invariance): it must stay below 1e-4 x its natural scale sum |d bias| (collected by `probe` from the fp64 run)."""
    failures = []
    for k, p in mod.named_parameters():
        if k.endswith(skip):
            continue
        try:
            if k.endswith(ZERO_GRADS):
                if probe is not None and p64[k].grad is not None:
                    assert_zero_grad("d" + k, p.grad, probe.scale(p64[k]))
                continue
            if p32[k].grad is None:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
                continue
            assert p.grad is not None, f"missing grad for {k}"
            alt = p32_alt[k].grad.cpu() if (p32_alt is not None and getattr(p32_alt[k], "grad", None) is not None) else None
            _calibrated("d" + k, p.grad, p32[k].grad, p64[k].grad, ref32_alt=alt)
        except AssertionError as e:
            failures.append(str(e).split("\n")[0])
    assert not failures, f"{len(failures)} parameter gradients out of tolerance:\n  " + "\n  ".join(failures)


# ------------------------------------------------------------------------------------------------
def test_gemm_layouts_and_epilogues(cuda):
    """A = I check with asymmetric B plus random NN / NT / TN products, batches, bias, relu, residual, split-K."""
    torch.manual_seed(0)
    M, N, K = 200, 77, 45
    A = torch.randn(M, K); Bm = torch.randn(K, N); bias = torch.randn(N); res = torch.randn(M, N)
    Ad, Bd = A.to(cuda), Bm.to(cuda)
    C = torch.empty(M, N, device=cuda)
    Fh._gemm(Ad, Bd, C, M=M, N=N, K=K, sam=K, sak=1, sbk=N, sbn=1, ldc=N)
    _assert_close("NN", C, A @ Bm, 1e-5)
    eye = torch.eye(64, device=cuda); asym = torch.arange(64 * 64, device=cuda, dtype=torch.float32).reshape(64, 64)
    C2 = torch.empty(64, 64, device=cuda)
    Fh._gemm(eye, asym, C2, M=64, N=64, K=64, sam=64, sak=1, sbk=64, sbn=1, ldc=64)
    assert torch.equal(C2, asym)
    # NT with bias + relu + residual
    W = torch.randn(N, K)
    Fh._gemm(Ad, W.to(cuda), C, M=M, N=N, K=K, sam=K, sak=1, sbk=1, sbn=K, ldc=N, bias=bias.to(cuda), bias_mode=1,
             residual=res.to(cuda), ldr=N, act=1)
    _assert_close("NT+bias+relu+res", C, torch.relu(A @ W.t() + bias) + res, 1e-5)
    # TN split-K (weight-gradient shape)
    R = 5000
    X = torch.randn(R, 40); dY = torch.randn(R, 24)
    dW = torch.zeros(24, 40, device=cuda)
    Fh._gemm(dY.to(cuda), X.to(cuda), dW, M=24, N=40, K=R, sam=1, sak=24, sbk=40, sbn=1, ldc=40, splitk=7)
    _assert_close("TN splitk", dW, dY.t() @ X, 1e-5)
    # batched (two batch dims) with per-row-block bias
    a = torch.randn(2, 3, 50, 20); b = torch.randn(2, 3, 20, 30); rb = torch.randn(2, 3, 5, 30)
    c = torch.empty(2, 3, 50, 30, device=cuda)
    Fh._gemm(a.to(cuda), b.to(cuda), c, M=50, N=30, K=20, sam=20, sak=1, sbk=30, sbn=1, ldc=30, nb0=2, nb1=3,
             sa0=3000, sa1=1000, sb0=1800, sb1=600, sc0=4500, sc1=1500, bias=rb.to(cuda), bias_mode=2, rows_per_bias=10,
             bias_ld=30, sbias0=450, sbias1=150)
    _assert_close("batched", c, a @ b + rb.repeat_interleave(10, dim=2), 1e-5)


def test_gemm_fast_path_matches_generic(cuda):
    """Aligned shapes take the tiled fast kernel; all four operand layouts, ragged M / N edges, K tails (zero-filled last
    tile: any K for row-contiguous operands, K % 4 == 0 for k-contiguous ones, else the generic kernel), split-K, epilogues."""
    torch.manual_seed(3)
    L = smml.lib()
    for (M, N, K) in [(300, 200, 64), (129, 65, 16), (1000, 64, 512), (77, 130, 32), (64, 16, 2500), (132, 72, 20),
                      (260, 128, 44), (96, 64, 18)]:
        A = torch.randn(M, K); At = A.t().contiguous(); Bm = torch.randn(K, N); Bt = Bm.t().contiguous()
        bias = torch.randn(N); res = torch.randn(M, N)
        ref = torch.relu(0.5 * (A @ Bm) + bias) + 2.0 * res
        for a_kc in (True, False):
            for b_kc in (True, False):
                if (not a_kc and M % 4) or (not b_kc and N % 4):
                    continue                      # unaligned row strides fall back to the generic kernel anyway
                C = torch.empty(M, N, device=cuda)
                Ad = (A if a_kc else At).to(cuda); Bd = (Bt if b_kc else Bm).to(cuda)
                Fh._gemm(Ad, Bd, C, M=M, N=N, K=K, sam=(K if a_kc else 1), sak=(1 if a_kc else M), sbk=(1 if b_kc else N),
                         sbn=(K if b_kc else 1), ldc=N, bias=bias.to(cuda), bias_mode=1, residual=res.to(cuda), ldr=N, act=1,
                         alpha=0.5, beta=2.0)
                _assert_close(f"fast {M}x{N}x{K} a_kc={a_kc} b_kc={b_kc}", C, ref, 1e-5)
        Cs = torch.zeros(M, N, device=cuda)
        Fh._gemm(A.to(cuda), Bt.to(cuda), Cs, M=M, N=N, K=K, sam=K, sak=1, sbk=1, sbn=K, ldc=N, splitk=3 if K >= 48 else 1,
                 accumulate=1)
        _assert_close(f"fast splitk {M}x{N}x{K}", Cs, A @ Bm, 1e-5)
    # same product through the generic kernel
    A = torch.randn(200, 64, device=cuda); Bm = torch.randn(64, 96, device=cuda)
    C1 = torch.empty(200, 96, device=cuda); C2 = torch.empty(200, 96, device=cuda)
    Fh._gemm(A, Bm, C1, M=200, N=96, K=64, sam=64, sak=1, sbk=96, sbn=1, ldc=96)
    L.smml_gemm_force_generic(1)
    try:
        Fh._gemm(A, Bm, C2, M=200, N=96, K=64, sam=64, sak=1, sbk=96, sbn=1, ldc=96)
    finally:
        L.smml_gemm_force_generic(0)
    _assert_close("fast vs generic", C1, C2, 1e-6)
    # the split-bf16 kernel (three bf16 terms per operand, six products) against the fp32-MFMA kernel: all four layouts,
    # ragged edges, a K tail, split-K
    for (M, N, K) in [(300, 200, 96), (260, 128, 44), (1024, 130, 2500)]:
        A = torch.randn(M, K); At = A.t().contiguous(); Bm = torch.randn(K, N); Bt = Bm.t().contiguous()
        ref = (A.double() @ Bm.double()).float()
        for a_kc in (True, False):
            for b_kc in (True, False):
                if (not a_kc and M % 4) or (not b_kc and N % 4) or ((a_kc or b_kc) and K % 4):
                    continue
                Ad = (A if a_kc else At).to(cuda); Bd = (Bt if b_kc else Bm).to(cuda)
                for mode in (1, 2):
                    L.smml_gemm_set_mode(mode)
                    try:
                        C = torch.zeros(M, N, device=cuda)
                        Fh._gemm(Ad, Bd, C, M=M, N=N, K=K, sam=(K if a_kc else 1), sak=(1 if a_kc else M),
                                 sbk=(1 if b_kc else N), sbn=(K if b_kc else 1), ldc=N, splitk=2 if K > 1000 else 1,
                                 accumulate=1 if K > 1000 else 0)
                    finally:
                        L.smml_gemm_set_mode(0)
                    _assert_close(f"mode {mode} {M}x{N}x{K} a_kc={a_kc} b_kc={b_kc}", C, ref, 2e-6)


def test_linear_layernorm_autograd(cuda):
    torch.manual_seed(1)
    x = torch.randn(3, 70, 96); w = torch.randn(50, 96) / 10; b = torch.randn(50); g = torch.randn(50); be = torch.randn(50)
    wy = torch.randn(3, 50)
    def run(dev, lin, ln_mean):
        xs, ws, bs, gs, bes = (t.clone().to(dev).requires_grad_() for t in (x, w, b, g, be))
        y = lin(xs, ws, bs)
        z = ln_mean(y, gs, bes)
        (z * wy.to(dev)).sum().backward()
        return [z, xs.grad, ws.grad, bs.grad, gs.grad, bes.grad]
    ref = run("cpu", lambda a, ww, bb: torch.relu(a @ ww.t() + bb),
              lambda y, gg, bb: torch.nn.functional.layer_norm(y, (50,), gg, bb).mean(dim=1))
    got = run(cuda, lambda a, ww, bb: Fh.linear(a, ww, bb, act=Fh.ACT_RELU), Fh.layer_norm_token_mean)
    for n, a, r in zip(("z", "dx", "dw", "db", "dgamma", "dbeta"), got, ref):
        _assert_close(n, a, r, 2e-5)


@pytest.mark.parametrize("rows", [1, 3, 16, 37, 1000])
def test_layernorm_128_columns(cuda, rows):
    """The 128-column LayerNorm backward has its own kernel (half-wave per row, four rows in flight): plain and
    token-mean forms (broadcast dy rows) at row counts that are not multiples of its 16-row blocks, vs torch on CPU."""
    torch.manual_seed(rows)
    x = torch.randn(2, rows, 128); g = torch.randn(128); be = torch.randn(128)
    wy = torch.randn(2, rows, 128); wm = torch.randn(2, 128)
    def run(dev, ln, ln_mean):
        xs, gs, bes = (t.clone().to(dev).requires_grad_() for t in (x, g, be))
        y = ln(xs, gs, bes)
        z = ln_mean(xs * 0.5 + 0.1, gs, bes)
        ((y * wy.to(dev)).sum() + (z * wm.to(dev)).sum()).backward()
        return [y, z, xs.grad, gs.grad, bes.grad]
    ref = run("cpu", lambda a, gg, bb: torch.nn.functional.layer_norm(a, (128,), gg, bb),
              lambda a, gg, bb: torch.nn.functional.layer_norm(a, (128,), gg, bb).mean(dim=1))
    got = run(cuda, Fh.layer_norm, Fh.layer_norm_token_mean)
    for n, a, r in zip(("y", "z", "dx", "dgamma", "dbeta"), got, ref):
        _assert_close(n, a, r, 2e-5)


@pytest.mark.parametrize("B,Hh,Ww", [(2, 12, 12), (1, 20, 20), (2, 14, 23), (1, 38, 38), (1, 6, 6), (2, 5, 9), (1, 31, 7)])
def test_deform2d_vs_oracle(cuda, B, Hh, Ww):
    """Small grids incl. non-square, N not a multiple of 128, J not a multiple of 32 (ragged tiles)."""
    C, N = 128, Hh * Ww
    tag = f"d2d:{B}:{Hh}:{Ww}"
    mod = smml.DeformCrossAttention2D(dim=C, dropout=0.1, grid_hw=(Hh, Ww))
    params = params_for(mod, 7, tag)
    mod = _load(mod, params, cuda)
    x1 = synth.normal((B, C, N), 7, tag + ":x1"); x2 = synth.normal((B, C, N), 7, tag + ":x2")
    w_out = synth.normal((B, C, N), 7, tag + ":wo")
    # HIP first: the decisions it takes (sampler cells, ReLU masks) are imposed on both oracle runs
    ad, bd = x1.to(cuda).requires_grad_(), x2.to(cuda).requires_grad_()
    with decision_tap() as tap:
        o, vg = mod(ad, bd, return_vgrid=True)
    w_vg = synth.normal(tuple(vg.shape), 7, tag + ":wvg")
    ((o * w_out.to(cuda)).sum() + (vg * w_vg.to(cuda)).sum()).backward()
    # oracle, fp32 (the reference's arithmetic) and fp64 (truth for the tolerance calibration)
    run = {}
    with cpb_probe() as probe:
        for dt in (torch.float32, torch.float64):
            pref = {k: v.clone().to(dt).requires_grad_() for k, v in params.items()}
            a, b = x1.clone().to(dt).requires_grad_(), x2.clone().to(dt).requires_grad_()
            odeform.DECISIONS = tap.decisions()
            o_ref, vg_ref = deform_cross_attention_2d(a, b, pref, grid_hw=(Hh, Ww))
            assert not odeform.DECISIONS
            ((o_ref * w_out.to(dt)).sum() + (vg_ref * w_vg.to(dt)).sum()).backward()
            run[dt] = (o_ref, vg_ref, a.grad, b.grad, pref)
    r32, r64 = run[torch.float32], run[torch.float64]
    for name, got, i in (("out", o, 0), ("vgrid", vg, 1), ("dx1", ad.grad, 2), ("dx2", bd.grad, 3)):
        _calibrated(name, got, r32[i], r64[i])
    _compare_param_grads(mod, r32[4], r64[4], probe=probe)
    # integer path: corners of the kernel's own sample positions, bit-exact against the oracle's formula
    th, tw = vg.shape[-2:]
    vgc = vg.detach().cpu()
    vsx = (2.0 * vgc[:, 0] / max(th - 1, 1) - 1.0).reshape(B * 8, -1)
    vsy = (2.0 * vgc[:, 1] / max(tw - 1, 1) - 1.0).reshape(B * 8, -1)
    _, _, corners = sample_positions(vsx, vsy, Ww, Hh)
    vs_dev = torch.stack((vsx, vsy), dim=-1).to(cuda)
    cx, cy, cm = Fh.bilinear_corners(vs_dev, Hh, Ww, 2)
    assert torch.equal(cx.cpu().long(), torch.stack([c[0] for c in corners], -1).reshape(-1, 4))
    assert torch.equal(cy.cpu().long(), torch.stack([c[1] for c in corners], -1).reshape(-1, 4))
    assert torch.equal(cm.cpu().bool(), torch.stack([c[3] for c in corners], -1).reshape(-1, 4))


def test_deform2d_train_mode_dropout(cuda):
    """train(): nn.Dropout(0.1) on the attention probabilities.  The kernels' counter-based mask for the drawn seed is
    exported through the C-ABI and fed to the oracle as an explicit mask; forward and gradients must then agree."""
    B, Hh, Ww, C = 2, 20, 20, 128
    N = Hh * Ww
    tag = "d2d:drop"
    mod = smml.DeformCrossAttention2D(dim=C, dropout=0.1, grid_hw=(Hh, Ww))
    params = params_for(mod, 13, tag)
    mod.load_state_dict(params)
    mod = mod.to(cuda).train()
    x1 = synth.normal((B, C, N), 13, tag + ":x1"); x2 = synth.normal((B, C, N), 13, tag + ":x2")
    w_out = synth.normal((B, C, N), 13, tag + ":wo")
    torch.manual_seed(1234)
    ad, bd = x1.to(cuda).requires_grad_(), x2.to(cuda).requires_grad_()
    with decision_tap() as tap:
        o, vg = mod(ad, bd, return_vgrid=True)
    (o * w_out.to(cuda)).sum().backward()
    J = vg.shape[-1] * vg.shape[-2]
    keep = Fh.deform_attention_dropout_mask(B, N, J, 8, 0.1, mod.last_dropout_seed, cuda).cpu()
    frac = float(keep.mean())
    assert 0.88 < frac < 0.92, f"keep fraction {frac}"
    run = {}
    with cpb_probe() as probe:
        for dt in (torch.float32, torch.float64):
            pref = {k: v.clone().to(dt).requires_grad_() for k, v in params.items()}
            a, b = x1.clone().to(dt).requires_grad_(), x2.clone().to(dt).requires_grad_()
            odeform.DECISIONS = tap.decisions()
            o_ref, vg_ref = deform_cross_attention_2d(a, b, pref, grid_hw=(Hh, Ww), attn_keep=keep, dropout_p=0.1)
            (o_ref * w_out.to(dt)).sum().backward()
            run[dt] = (o_ref, a.grad, b.grad, pref)
    r32, r64 = run[torch.float32], run[torch.float64]
    for name, got, i in (("out", o, 0), ("dx1", ad.grad, 1), ("dx2", bd.grad, 2)):
        _calibrated(name, got, r32[i], r64[i])
    _compare_param_grads(mod, r32[3], r64[3], probe=probe)
    # a second forward draws a new seed (different mask), eval() switches dropout off
    o2 = mod(ad.detach(), bd.detach())
    assert not torch.equal(o2, o.detach())
    mod.eval()
    o3, o4 = mod(ad.detach(), bd.detach()), mod(ad.detach(), bd.detach())
    assert torch.equal(o3, o4)


def test_deform2d_golden_reference_grid(cuda):
    """N = 2500 (50 x 50), dim 128: against outputs of the reference itself."""
    g = Golden("deform2d_ref50")
    B, C, N = 2, 128, 2500
    mod = smml.DeformCrossAttention2D(dim=C, dim_head=64, heads=8, dropout=0.1, downsample_factor=4, offset_scale=4,
                                      offset_groups=8, offset_kernel_size=6)
    mod = _load(mod, params_for(mod, 42, "deform2d"), cuda)
    x1 = synth.normal((B, C, N), 42, "deform2d:x1").to(cuda).requires_grad_()
    x2 = synth.normal((B, C, N), 42, "deform2d:x2").to(cuda).requires_grad_()
    w_out = synth.normal((B, C, N), 42, "deform2d:wout").to(cuda)
    w_vg = synth.normal((B * 8, 2, 12, 12), 42, "deform2d:wvg").to(cuda)
    out, vgrid = mod(x1, x2, return_vgrid=True)
    loss = (out * w_out).sum() + (vgrid * w_vg).sum()
    loss.backward()
    g.check("out", out); g.check("vgrid", vgrid); g.check("dx1", x1.grad); g.check("dx2", x2.grad)
    assert abs(loss.item() - g.scalar("loss")) <= 1e-4 * abs(g.scalar("loss"))
    for k, p in mod.named_parameters():
        if k.endswith(ZERO_GRADS):
            assert_zero_grad(f"{g.name}:d{k}", p.grad, g.scalar("natural:" + k))
        else:
            g.check("grad:" + k, p.grad, what="d" + k)
    # integer path on the REFERENCE's vgrid: bit-exact corners / masks
    vgf = torch.from_numpy(g.array("vgrid_full"))
    vs = (2.0 * vgf / 11.0 - 1.0)
    vs_dev = torch.stack((vs[:, 0].reshape(16, 144), vs[:, 1].reshape(16, 144)), dim=-1).to(cuda)
    cx, cy, cm = Fh.bilinear_corners(vs_dev, 50, 50, 2)
    assert np.array_equal(cx.cpu().numpy().reshape(16, 144, 4), g.array("corner_x"))
    assert np.array_equal(cy.cpu().numpy().reshape(16, 144, 4), g.array("corner_y"))
    assert np.array_equal(cm.cpu().numpy().astype(bool).reshape(16, 144, 4), g.array("corner_mask"))


@pytest.mark.parametrize("B,n", [(2, 5), (1, 64), (3, 129), (2, 300)])
def test_deform1d_vs_oracle_lengths(cuda, B, n):
    """1-D module at lengths outside the golden set (a single sampled key, exact tile multiples, ragged): two heads per
    offset group, degenerate-axis sampling; outputs and gradients against the oracle (fp64-calibrated tolerance)."""
    from oracle.deform import deform_cross_attention_1d
    C = 128
    tag = f"d1d:{B}:{n}"
    mod = smml.DeformCrossAttention1D(dim=C, downsample_factor=4, offset_scale=2, offset_kernel_size=6)
    params = params_for(mod, 9, tag)
    mod = _load(mod, params, cuda)
    x1 = synth.normal((B, C, n), 9, tag + ":x1"); x2 = synth.normal((B, C, n), 9, tag + ":x2")
    wo = synth.normal((B, C, n), 9, tag + ":wo")
    ad, bd = x1.to(cuda).requires_grad_(), x2.to(cuda).requires_grad_()
    with decision_tap() as tap:
        o, vg = mod(ad, bd, return_vgrid=True)
    w_vg = synth.normal(tuple(vg.shape), 9, tag + ":wvg")
    ((o * wo.to(cuda)).sum() + (vg * w_vg.to(cuda)).sum()).backward()
    run = {}
    with cpb_probe() as probe:
        for dt in (torch.float32, torch.float64):
            pref = {k: v.clone().to(dt).requires_grad_() for k, v in params.items()}
            a, b = x1.clone().to(dt).requires_grad_(), x2.clone().to(dt).requires_grad_()
            odeform.DECISIONS = tap.decisions()
            o_ref, vg_ref = deform_cross_attention_1d(a, b, pref, downsample_factor=4, offset_scale=2, offset_kernel_size=6)
            ((o_ref * wo.to(dt)).sum() + (vg_ref * w_vg.to(dt)).sum()).backward()
            run[dt] = (o_ref, vg_ref, a.grad, b.grad, pref)
    r32, r64 = run[torch.float32], run[torch.float64]
    for name, got, i in (("out", o, 0), ("vgrid", vg, 1), ("dx1", ad.grad, 2), ("dx2", bd.grad, 3)):
        _calibrated(name, got, r32[i], r64[i])
    _compare_param_grads(mod, r32[4], r64[4], probe=probe)


@pytest.mark.parametrize("tag,B,n", [("deform1d_n37", 2, 37), ("deform1d_n40", 2, 40), ("deform1d_n2501", 1, 2501)])
def test_deform1d_golden(cuda, tag, B, n):
    g = Golden(tag)
    C = 128
    mod = smml.DeformCrossAttention1D(dim=C, downsample_factor=4, offset_scale=2, offset_kernel_size=6)
    mod = _load(mod, params_for(mod, 42, tag), cuda)
    x1 = synth.normal((B, C, n), 42, tag + ":x1").to(cuda).requires_grad_()
    x2 = synth.normal((B, C, n), 42, tag + ":x2").to(cuda).requires_grad_()
    w_out = synth.normal((B, C, n), 42, tag + ":wout").to(cuda)
    out, vgrid = mod(x1, x2, return_vgrid=True)
    w_vg = synth.normal(tuple(vgrid.shape), 42, tag + ":wvg").to(cuda)
    ((out * w_out).sum() + (vgrid * w_vg).sum()).backward()
    g.check("out", out); g.check("vgrid", vgrid); g.check("dx1", x1.grad); g.check("dx2", x2.grad)
    for k, p in mod.named_parameters():
        if k.endswith(ZERO_GRADS):
            assert_zero_grad(f"{g.name}:d{k}", p.grad, g.scalar("natural:" + k))
        else:
            g.check("grad:" + k, p.grad, what="d" + k)


@pytest.mark.parametrize("tag,B,n", [("deform1d_rawdist_n40", 2, 40), ("deform1d_rawdist_n300", 1, 300)])
@pytest.mark.parametrize("mode", [None, "bf16"])
def test_deform1d_raw_distance_golden(cuda, tag, B, n, mode):
    """DeformCrossAttention1D(cpb_log_distance=False) - the bias MLP reads the raw offset - against the reference's own output (fp32-grade path, the
    golden's tolerance) and, in the 16-bit compute mode, against the fp64 oracle with the kernels' decisions imposed (that mode's tolerances)."""
    C = 128
    mod = smml.DeformCrossAttention1D(dim=C, downsample_factor=4, offset_scale=2, offset_kernel_size=6, cpb_log_distance=False, compute_dtype=mode)
    params = params_for(mod, 42, tag)
    mod = _load(mod, params, cuda)
    x1h = synth.normal((B, C, n), 42, tag + ":x1"); x2h = synth.normal((B, C, n), 42, tag + ":x2")
    x1 = x1h.to(cuda).requires_grad_(); x2 = x2h.to(cuda).requires_grad_()
    w_out = synth.normal((B, C, n), 42, tag + ":wout")
    with decision_tap() as tap:
        out, vgrid = mod(x1, x2, return_vgrid=True)
    w_vg = synth.normal(tuple(vgrid.shape), 42, tag + ":wvg")
    ((out * w_out.to(cuda)).sum() + (vgrid * w_vg.to(cuda)).sum()).backward()
    if mode is None:
        g = Golden(tag)
        g.check("out", out); g.check("vgrid", vgrid); g.check("dx1", x1.grad); g.check("dx2", x2.grad)
        for k, p in mod.named_parameters():
            if k.endswith(ZERO_GRADS):
                assert_zero_grad(f"{g.name}:d{k}", p.grad, g.scalar("natural:" + k))
            else:
                g.check("grad:" + k, p.grad, what="d" + k)
        return
    dt = torch.float64
    pref = {k: v.clone().to(dt).requires_grad_() for k, v in params.items()}
    a, b = x1h.clone().to(dt).requires_grad_(), x2h.clone().to(dt).requires_grad_()
    odeform.DECISIONS = tap.decisions()
    o_ref, vg_ref = deform_cross_attention_1d(a, b, pref, downsample_factor=4, offset_scale=2, offset_kernel_size=6, cpb_log_distance=False)
    ((o_ref * w_out.to(dt)).sum() + (vg_ref * w_vg.to(dt)).sum()).backward()
    _assert_close(tag + " bf16 out", out, o_ref, 1.5e-2); _assert_close(tag + " bf16 dx1", x1.grad, a.grad, 3e-2); _assert_close(tag + " bf16 dx2", x2.grad, b.grad, 3e-2)
    for k, p in mod.named_parameters():
        if k.endswith(ZERO_GRADS) or pref[k].grad is None:
            continue
        _assert_close(f"{tag} bf16 d{k}", p.grad, pref[k].grad, 6e-2 if "rel_pos_bias" in k else 3e-2)


def test_batch_loss_vs_oracle(cuda):
    B = 4
    g = Golden("batchloss_b4")
    omic = synth.normal((B, 50, 16), 42, "bl:omic").to(cuda).requires_grad_()
    vgrid = synth.normal((B * 8, 2, 3, 3), 42, "bl:vgrid").to(cuda).requires_grad_()
    out = smml.BatchLoss(B, 1)(omic, vgrid)
    out.sum().backward()
    g.check("out", out); g.check("domic", omic.grad); g.check("dvgrid", vgrid.grad)
    # tile hint: a tile made by tile_tokens is reduced to its vector before the Gram - same loss, same gradient
    vec = synth.normal((B, 16), 3, "bl:vec")
    vg = synth.normal((B * 8, 2, 3, 3), 3, "bl:vg2").to(cuda)
    res = []
    for hint in (True, False):
        v = vec.clone().to(cuda).requires_grad_()
        l = smml.BatchLoss(B, 1, use_tile_hint=hint)(Fh.tile_tokens(v, 50), vg).sum()
        l.backward()
        res.append((l.detach(), v.grad))
    _assert_close("tile hint loss", res[0][0], res[1][0], 1e-5); _assert_close("tile hint grad", res[0][1], res[1][1], 1e-4)
    # long contraction (K = N*C as in the model) against the oracle
    o2 = synth.normal((8, 3000, 128), 1, "bl2:omic"); v2 = synth.normal((64, 2, 12, 12), 1, "bl2:vg")
    ref = batch_loss(o2, v2, 8)
    got = smml.BatchLoss(8, 1)(o2.to(cuda), v2.to(cuda))
    _assert_close("batchloss long-K", got, ref)


def test_full_model_golden_reference_grid(cuda):
    """DeformPathomicNet (two DeformCrossTransMIL branches) + BatchLoss + CE at N = 2500 against the reference."""
    g = Golden("pathomic_ref50")
    net = smml.DeformPathomicNet(pathomic_args())
    net = _load(net, params_for(net, 42, "pathomic"), cuda)
    B = 2
    x_path = synth.bag(B, 2500, 1024, 42, "pathomic:bag").to(cuda)
    x_t = synth.normal((B, 59), 42, "pathomic:tumor").to(cuda)
    x_i = synth.normal((B, 361), 42, "pathomic:immune").to(cuda)
    feats, vt, vi, lg, _, _, _ = net(x_path=x_path, x_omic=None, x_omic_tumor=x_t, x_omic_immune=x_i)
    bl = smml.BatchLoss(B, 1)
    l_t, l_i = bl(lg[3], lg[4]), bl(lg[5], lg[6])
    label = torch.tensor([1, 3], device=cuda)
    loss = torch.nn.functional.cross_entropy(lg[2], label) + 0.5 * l_t.sum() + 0.5 * l_i.sum()
    loss.backward()
    g.check("features", feats); g.check("vec_t", vt); g.check("vec_i", vi); g.check("haz", lg[2])
    g.check("haz_t", lg[0]); g.check("haz_i", lg[1])
    g.check("vgrid_t", lg[4]); g.check("vgrid_i", lg[6]); g.check("batchloss_t", l_t); g.check("batchloss_i", l_i)
    g.check("omic_t_row0", lg[3][:, 0])
    assert abs(loss.item() - g.scalar("loss")) <= 1e-4 * abs(g.scalar("loss"))
    with_grad = {k for k, p in net.named_parameters() if p.grad is not None}
    assert with_grad == {k[5:] for k in g.keys("grad:")}, "set of parameters receiving a gradient differs"
    for k, p in net.named_parameters():
        if p.grad is not None and k.endswith(ZERO_GRADS):
            assert_zero_grad(f"{g.name}:d{k}", p.grad, g.scalar("natural:" + k))
        elif p.grad is not None:
            g.check("grad:" + k, p.grad, what="d" + k)


def test_mil_branch_larger_grid_vs_oracle(cuda):
    """One DeformCrossTransMIL branch on a 64 x 64 grid with 512-wide bag features (BASELINE config 3 shape)."""
    args = pathomic_args(input_path_dim=512, return_vgrid=True)
    mil = smml.DeformCrossTransMIL(args)
    params = params_for(mil, 3, "mil64")
    mil = _load(mil, params, cuda)
    B, S = 1, 64
    path = synth.bag(B, S * S, 512, 3, "mil64:bag"); omic = torch.relu(synth.normal((B, 128), 3, "mil64:omic"))
    with decision_tap() as tap:
        enc, logits, _, omic_t, vg = mil(path.to(cuda), omic.to(cuda))
    (enc.sum() + (logits * logits).sum() + vg.pow(2).sum() * 1e-3).backward()
    run = {}
    with cpb_probe() as probe:
        for dt in (torch.float32, torch.float64):
            pref = {k: v.clone().to(dt).requires_grad_() for k, v in params.items()}
            odeform.DECISIONS = tap.decisions()
            enc_r, log_r, _, vg_r = deform_cross_trans_mil(path.to(dt), omic.to(dt), pref, grid_hw=(S, S))
            (enc_r.sum() + (log_r * log_r).sum() + vg_r.pow(2).sum() * 1e-3).backward()
            run[dt] = (enc_r, log_r, vg_r, pref)
    r32, r64 = run[torch.float32], run[torch.float64]
    for name, got, i in (("encoded", enc, 0), ("logits", logits, 1), ("vgrid", vg, 2)):
        _calibrated(name, got, r32[i], r64[i])
    assert omic_t.shape == (B, S * S, 128) and torch.equal(omic_t[0, 17].cpu(), omic[0])
    _compare_param_grads(mil, r32[3], r64[3], skip=("cls_token",), probe=probe)


def test_size_independent_properties_full_size(cuda):
    """N = 10 000 (100 x 100 grid, 625 keys): properties that need no oracle run -
    (1) attention rows are convex combinations: with v = 1 the core returns exactly-normalised ones;
    (2) the core is linear in v; (3) permuting bags permutes outputs (whole-bag independence)."""
    torch.manual_seed(0)
    B, N, J, H = 2, 10000, 625, 8
    q = torch.randn(B, N, 512, device=cuda) * 0.3; k = torch.randn(B, J, 512, device=cuda) * 0.3
    v = torch.randn(B, J, 512, device=cuda); vs = torch.rand(B * 8, J, 2, device=cuda) * 2 - 1
    mod = smml.DeformCrossAttention2D(dim=128).to(cuda)
    cp = [t.detach() for t in mod.rel_pos_bias.tensors()]
    gq = smml.deform_attention._grid_queries_2d(100, 100, cuda)
    f = lambda vv, qq=q, kk=k, ss=vs: Fh.deform_attention(qq, kk, vv, ss, gq, *cp, heads=H, groups=8, scale=0.125)
    ones = f(torch.ones_like(v))
    assert float((ones - 1).abs().max()) < 1e-5
    o1, o2 = f(v), f(2 * v + 1)
    assert float((o2 - (2 * o1 + 1)).abs().max()) < 1e-4
    perm = torch.tensor([1, 0], device=cuda)
    vs_p = vs.reshape(B, 8, J, 2)[perm].reshape(B * 8, J, 2)
    o_p = Fh.deform_attention(q[perm], k[perm], v[perm], vs_p, gq, *cp, heads=H, groups=8, scale=0.125)
    assert torch.equal(o_p, o1[perm])


def test_fails_loudly_on_cpu_tensors(cuda):
    mod = smml.DeformCrossAttention2D(dim=128).eval()
    with pytest.raises(RuntimeError):
        mod(torch.randn(1, 128, 144), torch.randn(1, 128, 144))


def _core_reference(q, k, v, vs, gq, w1, b1, w2, b2, w3, b3, heads, groups, scale, keep=None, keep_scale=1.0, masks=None):
    """dropout(softmax(scale q k^T + CPB(gq - vs))) v in plain torch (any dtype / device): the fused core's contract."""
    B, N, _ = q.shape
    J, PD, o = k.shape[1], vs.shape[-1], heads // groups
    pos = gq[None, :, None, :] - vs.view(B * groups, 1, J, PD)
    p = torch.sign(pos) * torch.log(pos.abs() + 1)
    if masks is None:
        h2 = torch.relu(torch.relu(p @ w1.T + b1) @ w2.T + b2)
    else:                            # the kernels' own ReLU decisions (helpers.Decisions), bool [(B G), N, J, 32] each
        h2 = (((p @ w1.T + b1) * masks[0].to(p.dtype)) @ w2.T + b2) * masks[1].to(p.dtype)
    bias = (h2 @ w3.T + b3).view(B, groups, N, J, o).permute(0, 1, 4, 2, 3).reshape(B, heads, N, J)
    d = q.shape[-1] // heads
    qh = q.view(B, N, heads, d).permute(0, 2, 1, 3) * scale
    kh = k.view(B, J, heads, d).permute(0, 2, 1, 3)
    vh = v.view(B, J, heads, d).permute(0, 2, 1, 3)
    attn = torch.softmax(qh @ kh.transpose(-1, -2) + bias, dim=-1)
    if keep is not None:
        attn = attn * (keep.to(attn.dtype) * keep_scale)
    return (attn @ vh).permute(0, 2, 1, 3).reshape(B, N, heads * d)


def test_fused_core_random_shapes(cuda):
    """The fused attention core (forward + all three backward passes) on random ragged shapes: N and J that are not
    multiples of the 32 / 128-wide tiles, one or two heads per offset group, 1-D and 2-D positions, with and without
    dropout - against a plain torch evaluation in fp64 (tolerance calibrated by the same evaluation in fp32)."""
    import os
    nrand = int(os.environ.get("SMML_FUZZ_CASES", "14"))          # soak runs: SMML_FUZZ_CASES=100 SMML_FUZZ_SEED=7 pytest -k random_shapes
    gen = torch.Generator().manual_seed(int(os.environ.get("SMML_FUZZ_SEED", "1234")))
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen))
    forced = [(1, 1, 65, 8, 1, 0.0), (2, 33, 5, 4, 2, 0.25), (1, 129, 33, 8, 2, 0.0)]   # single query (every wave but one past the
    for case in range(nrand + len(forced)):                                            # bag end), ragged tiles
        B, N, J = ri(1, 3), ri(1, 300), ri(1, 90)
        groups = (4, 8)[ri(0, 1)]
        heads, PD, p_drop = 8, ri(1, 2), (0.0, 0.25)[ri(0, 1)]
        if case >= nrand:
            B, N, J, groups, PD, p_drop = forced[case - nrand]
        rn = lambda *s: torch.randn(*s, generator=gen)
        t = dict(q=rn(B, N, 512) * 0.4, k=rn(B, J, 512) * 0.4, v=rn(B, J, 512), vs=torch.rand(B * groups, J, PD, generator=gen) * 2.4 - 1.2,
                 gq=torch.rand(N, PD, generator=gen) * 2 - 1, w1=rn(32, PD) * 0.7, b1=rn(32) * 0.3, w2=rn(32, 32) * 0.25, b2=rn(32) * 0.2,
                 w3=rn(heads // groups, 32) * 0.3, b3=rn(heads // groups) * 0.1)
        wo = rn(B, N, 512)
        dev = {n: x.to(cuda).requires_grad_() for n, x in t.items()}
        seed = 17 + case
        smml.functional.DECISION_TAP = tapped = []
        try:
            out = Fh.deform_attention(*(dev[n] for n in ("q", "k", "v", "vs", "gq", "w1", "b1", "w2", "b2", "w3", "b3")), heads=heads,
                                      groups=groups, scale=0.125, dropout_p=p_drop, dropout_seed=seed)
        finally:
            smml.functional.DECISION_TAP = None
        (out * wo.to(cuda)).sum().backward()
        keep = Fh.deform_attention_dropout_mask(B, N, J, heads, p_drop, seed, cuda) if p_drop else None
        # the ReLU decisions the kernels took (layer 1 exported, layer 2 as saved for the backward), imposed on both torch evaluations
        m1, m2 = helpers.decisions_of(tapped[0], cuda)          # either kernel family: per-pair MLP (saved bits) or linear regions
        refs = {}
        for dt in (torch.float32, torch.float64):
            r = {n: x.to(cuda, dt).requires_grad_() for n, x in t.items()}
            o = _core_reference(*(r[n] for n in ("q", "k", "v", "vs", "gq", "w1", "b1", "w2", "b2", "w3", "b3")), heads, groups, 0.125,
                                keep, 1.0 / (1.0 - p_drop), masks=(m1, m2))
            (o * wo.to(cuda, dt)).sum().backward()
            refs[dt] = (o, r)
        with torch.no_grad():       # the exported decisions against the exact (fp64) pre-activations: they may only differ at rounding level
            r64 = refs[torch.float64][1]
            pos = r64["gq"][None, :, None, :] - r64["vs"].view(B * groups, 1, J, PD)
            x1 = (torch.sign(pos) * torch.log(pos.abs() + 1)) @ r64["w1"].T + r64["b1"]
            x2 = torch.relu(x1) @ r64["w2"].T + r64["b2"]
            for nm, x, m in (("layer 1", x1, m1), ("layer 2", x2, m2)):
                bad = x[(x > 0) != m].abs()
                assert bad.numel() == 0 or float(bad.max()) < 2e-6, f"case {case}: a {nm} decision with |pre-activation| {float(bad.max()):.2e} differs from fp64"
        tag = f"case {case}: B={B} N={N} J={J} G={groups} PD={PD} p={p_drop}"
        _calibrated(tag + " out", out, refs[torch.float32][0], refs[torch.float64][0])
        for n in t:
            if n in ("gq", "b3"):
                continue                                   # the query grid is a constant; d b3 is identically zero (softmax shift)
            g, g32, g64 = dev[n].grad, refs[torch.float32][1][n].grad, refs[torch.float64][1][n].grad
            if float(g64.abs().max()) < 1e-9:               # identically zero in exact arithmetic (one key: dS = P (dP - delta) = 0);
                # what is left is the rounding of dP - delta, i.e. ~1e-6 of |dP| ~ |v| |d out| d = O(100)
                assert float(g.abs().max()) < 1e-3, f"{tag} d{n}: expected ~0, got {float(g.abs().max()):.3e}"
                continue
            # every gradient under the plain rules (max-norm max(1e-4, 2 x noise), l2 max(1e-4, 1.5 x noise)): with the kernels' ReLU
            # decisions imposed on both torch evaluations there is no undecidable-mask case any more
            _calibrated(tag + " d" + n, g, g32, g64)


def test_fused_core_gradients_are_run_to_run_identical(cuda):
    """Every gradient of the fused attention core is reduced in a fixed order (per-workgroup / per-wave slabs, no float atomics):
    two runs on the same inputs give the same bits - d vs included (it used to go through LDS + HBM atomics)."""
    gen = torch.Generator().manual_seed(21)
    B, N, J, heads, groups, PD = 2, 700, 150, 8, 4, 2
    rn = lambda *s: torch.randn(*s, generator=gen)
    t = dict(q=rn(B, N, 512) * 0.4, k=rn(B, J, 512) * 0.4, v=rn(B, J, 512), vs=torch.rand(B * groups, J, PD, generator=gen) * 2.4 - 1.2,
             gq=torch.rand(N, PD, generator=gen) * 2 - 1, w1=rn(32, PD) * 0.7, b1=rn(32) * 0.3, w2=rn(32, 32) * 0.25, b2=rn(32) * 0.2,
             w3=rn(heads // groups, 32) * 0.3, b3=rn(heads // groups) * 0.1)
    wo = rn(B, N, 512).to(cuda)
    names = ("q", "k", "v", "vs", "gq", "w1", "b1", "w2", "b2", "w3", "b3")
    runs = []
    for _ in range(2):
        dev = {n: x.to(cuda).requires_grad_(n != "gq") for n, x in t.items()}
        out = Fh.deform_attention(*(dev[n] for n in names), heads=heads, groups=groups, scale=0.125, dropout_p=0.1, dropout_seed=5)
        (out * wo).sum().backward()
        torch.cuda.synchronize()
        runs.append({n: dev[n].grad.clone() for n in names if n != "gq"} | {"out": out.detach().clone()})
    for n in runs[0]:
        assert torch.equal(runs[0][n], runs[1][n]), f"{n} differs between two runs"


def test_saved_relu_masks_match_reference(cuda):
    """What the fused forward keeps of the position bias's hidden layer for its backward: one bit per unit
    (include/smml.h: relu_masks [B, H, J, 2, nst] uint16, hidden channel acc_row(r, half) of lane (query, half) at bit
    (13 + r) % 16).  Decoded, the bits must equal [W2 relu(W1 p + b1) + b2 > 0] of a torch fp64 evaluation wherever that
    pre-activation is not within fp32 rounding of zero; padding columns are ignored."""
    capi = smml._capi
    gen = torch.Generator().manual_seed(5)
    B, N, J, H, G, PD = 2, 150, 37, 8, 4, 2
    rn = lambda *s: torch.randn(*s, generator=gen)
    t = dict(q=rn(B, N, 512) * 0.4, k=rn(B, J, 512) * 0.4, v=rn(B, J, 512), vs=torch.rand(B * G, J, PD, generator=gen) * 2.4 - 1.2,
             gq=torch.rand(N, PD, generator=gen) * 2 - 1, w1=rn(32, PD) * 0.7, b1=rn(32) * 0.3, w2=rn(32, 32) * 0.25, b2=rn(32) * 0.2,
             w3=rn(H // G, 32) * 0.3, b3=rn(H // G) * 0.1)
    d = {n: x.to(cuda).contiguous() for n, x in t.items()}
    L = capi.lib()
    nst = L.smml_deform_attn_nst(N)
    assert nst % 128 == 0 and nst >= N
    out = torch.empty(B, N, 512, device=cuda)
    lse = torch.empty(B, H, N, device=cuda)
    logits = torch.empty(B, H, nst // 32, J, 32, device=cuda)                                   # per-tile storage (include/smml.h)
    masks = torch.zeros(B, H, nst // 32, J, 2, 32, device=cuda, dtype=torch.int16)
    capi.check(L.smml_deform_attn_fwd_f32(*(capi.fptr(d[n]) for n in ("q", "k", "v", "vs", "gq", "w1", "b1", "w2", "b2", "w3", "b3")),
                                          capi.fptr(out), capi.fptr(lse), capi.fptr(logits), capi.ptr(masks), B, N, J, H, G, PD,
                                          0.125, 0.0, 0, None, None, capi.stream(), None), "deform_attn_fwd")
    torch.cuda.synchronize()
    r = {n: x.to(cuda, torch.float64) for n, x in t.items()}
    pos = r["gq"][None, :, None, :] - r["vs"].view(B * G, 1, J, PD)
    x2 = torch.relu((torch.sign(pos) * torch.log(pos.abs() + 1)) @ r["w1"].T + r["b1"]) @ r["w2"].T + r["b2"]   # [(B G), N, J, 32]
    x2 = x2.view(B, G, N, J, 32)
    bits = Fh.relu_masks_rows(masks).to(torch.int32) & 0xFFFF                                # [B, H, J, 2, nst]
    o = H // G
    checked = wrong = 0
    for half in range(2):
        for reg in range(16):
            ch = (reg & 3) + 8 * (reg >> 2) + 4 * half                                      # acc_row(reg, half)
            got = ((bits[:, :, :, half, :N] >> ((13 + reg) % 16)) & 1).bool()               # [B, H, J, N]
            ref = x2[..., ch].permute(0, 1, 3, 2).repeat_interleave(o, dim=1)               # [B, H, J, N] (heads of a group share layers 1-2)
            sure = ref.abs() > 1e-5
            checked += int(sure.sum())
            wrong += int(((got != (ref > 0)) & sure).sum())
    assert checked > 0.99 * B * H * J * N * 32 and wrong == 0, f"{wrong} of {checked} saved ReLU decisions differ"


def test_saved_relu_masks_statistics_at_scale(cuda):
    """2.3e8 layer-2 ReLU decisions (3000 queries x 300 keys x 8 groups x 32 units) against a torch fp64 evaluation, beside the
    decisions of torch's own fp32 evaluation.  A decision can only differ where the fp64 pre-activation lies within fp32 rounding
    of zero, and a handful do (profiles/r02_split_terms.txt: 7 - 9 for every variant of the forward's chain, 6 for torch fp32).
    ONE such unit moves d W1 / d W2 by ~4e-4 of their norm (a sum of 7e6 random-sign terms), which is why the gradient-level l2
    rule of the large tests is a small-number statistic; this test bounds the cause itself: how many decisions differ and how far
    from zero those pre-activations are."""
    capi = smml._capi
    gen = torch.Generator().manual_seed(7)
    B, N, J, H, G, PD = 1, 3000, 300, 8, 8, 2
    rn = lambda *s: torch.randn(*s, generator=gen)
    t = dict(q=rn(B, N, 512) * 0.4, k=rn(B, J, 512) * 0.4, v=rn(B, J, 512), vs=torch.rand(B * G, J, PD, generator=gen) * 2.4 - 1.2,
             gq=torch.rand(N, PD, generator=gen) * 2 - 1, w1=rn(32, PD) * 0.7, b1=rn(32) * 0.3, w2=rn(32, 32) * 0.25, b2=rn(32) * 0.2,
             w3=rn(H // G, 32) * 0.3, b3=rn(H // G) * 0.1)
    d = {n: x.to(cuda).contiguous() for n, x in t.items()}
    L = capi.lib()
    nst = L.smml_deform_attn_nst(N)
    out = torch.empty(B, N, 512, device=cuda)
    lse = torch.empty(B, H, N, device=cuda)
    logits = torch.empty(B, H, nst // 32, J, 32, device=cuda)                                   # per-tile storage (include/smml.h)
    masks = torch.zeros(B, H, nst // 32, J, 2, 32, device=cuda, dtype=torch.int16)
    capi.check(L.smml_deform_attn_fwd_f32(*(capi.fptr(d[n]) for n in ("q", "k", "v", "vs", "gq", "w1", "b1", "w2", "b2", "w3", "b3")),
                                          capi.fptr(out), capi.fptr(lse), capi.fptr(logits), capi.ptr(masks), B, N, J, H, G, PD,
                                          0.125, 0.0, 0, None, None, capi.stream(), None), "deform_attn_fwd")
    torch.cuda.synchronize()

    def pre(dt):
        r = {n: t[n].to(cuda, dt) for n in ("gq", "vs", "w1", "b1", "w2", "b2")}
        pos = r["gq"][None, :, None, :] - r["vs"].view(B * G, 1, J, PD)
        return (torch.relu((torch.sign(pos) * torch.log(pos.abs() + 1)) @ r["w1"].T + r["b1"]) @ r["w2"].T + r["b2"]).view(B, G, N, J, 32)
    x64 = pre(torch.float64)
    n32 = int(((pre(torch.float32) > 0) != (x64 > 0)).sum())
    bits = Fh.relu_masks_rows(masks).to(torch.int32) & 0xFFFF
    flipped = []
    for half in range(2):
        for reg in range(16):
            ch = (reg & 3) + 8 * (reg >> 2) + 4 * half
            got = ((bits[:, :, :, half, :N] >> ((13 + reg) % 16)) & 1).bool()               # [B, H, J, N]; o = 1: head = group
            ref = x64[..., ch].permute(0, 1, 3, 2)
            flipped.append(ref[got != (ref > 0)].abs())
    flipped = torch.cat(flipped)
    nflip, worst = int(flipped.numel()), (float(flipped.max()) if flipped.numel() else 0.0)
    helpers.record("layer-2 decisions differing from fp64 (count; fp32_noise column = torch fp32's count)", nflip, n32, 3 * n32 + 10, "count")
    helpers.record("largest |fp64 pre-activation| of a differing decision", worst, None, 1e-6, "abs")
    assert nflip <= 3 * n32 + 10, f"{nflip} decisions differ from fp64 (torch fp32: {n32})"
    assert worst < 1e-6, f"a decision with pre-activation {worst:.2e} differs: not a rounding-level tie"


def test_gemm_random_shapes(cuda):
    """smml_gemm_f32 on random shapes, operand layouts (k- or row-contiguous, padded leading dimensions that keep or break
    the 16-byte alignment of the tiled kernels), batch dimensions, split-K and epilogues, against torch in fp64."""
    gen = torch.Generator().manual_seed(99)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen))
    for case in range(40):
        M, N, K = ri(1, 300), ri(1, 200), ri(1, 260)
        nb0, nb1 = ri(1, 2), ri(1, 3)
        a_kc, b_kc = bool(ri(0, 1)), bool(ri(0, 1))
        pa, pb = (0, 4, 3)[ri(0, 2)], (0, 4, 1)[ri(0, 2)]                 # leading-dimension padding: aligned / unaligned
        A = torch.randn(nb0, nb1, M, K, generator=gen); Bm = torch.randn(nb0, nb1, K, N, generator=gen)
        lda = (K if a_kc else M) + pa; ldb = (K if b_kc else N) + pb
        As = torch.zeros(nb0, nb1, (M if a_kc else K), lda); As[..., :(K if a_kc else M)] = A if a_kc else A.transpose(-1, -2)
        Bs = torch.zeros(nb0, nb1, (N if b_kc else K), ldb); Bs[..., :(K if b_kc else N)] = Bm.transpose(-1, -2) if b_kc else Bm
        mode = ri(0, 3)                                                    # 0 plain, 1 bias+relu, 2 alpha/beta residual, 3 split-K accumulate
        alpha = 1.0 if mode in (0, 1, 3) else 0.7
        ref = alpha * (A.double() @ Bm.double())
        kw = dict(M=M, N=N, K=K, sam=(lda if a_kc else 1), sak=(1 if a_kc else lda), sbk=(1 if b_kc else ldb), sbn=(ldb if b_kc else 1),
                  ldc=N, nb0=nb0, nb1=nb1, sa0=As.stride(0), sa1=As.stride(1), sb0=Bs.stride(0), sb1=Bs.stride(1), sc0=nb1 * M * N,
                  sc1=M * N, alpha=alpha)
        C = torch.zeros(nb0, nb1, M, N, device=cuda)
        if mode == 1:
            bias = torch.randn(N, generator=gen)
            ref = torch.relu(ref + bias.double()); kw.update(bias=bias.to(cuda), bias_mode=1, act=1)
        elif mode == 2:
            res = torch.randn(nb0, nb1, M, N, generator=gen)
            ref = ref + 1.5 * res.double(); kw.update(residual=res.to(cuda), ldr=N, beta=1.5)
        elif mode == 3:
            kw.update(splitk=ri(2, 5), accumulate=1)
        Fh._gemm(As.to(cuda), Bs.to(cuda), C, **kw)
        _assert_close(f"gemm case {case}: {nb0}x{nb1} {M}x{N}x{K} a_kc={a_kc} b_kc={b_kc} pads {pa},{pb} mode {mode}", C, ref.float(), 1e-5)


def test_exported_layer1_decisions_match_fp64(cuda):
    """smml_deform_attn_relu1_masks - the layer-1 ReLU decisions the parity tests impose on the oracle - against a torch fp64
    evaluation: they may differ only where the fp64 pre-activation is within fp32 rounding of zero (1-D and 2-D positions, two heads
    per group, N / J not multiples of the tiles)."""
    gen = torch.Generator().manual_seed(11)
    rn = lambda *s: torch.randn(*s, generator=gen)
    for (B, N, J, G, PD) in [(2, 150, 37, 4, 2), (1, 700, 130, 8, 2), (3, 129, 64, 4, 1)]:
        vs = (torch.rand(B * G, J, PD, generator=gen) * 2.4 - 1.2).to(cuda)
        gq = (torch.rand(N, PD, generator=gen) * 2 - 1).to(cuda)
        w1, b1 = (rn(32, PD) * 0.7).to(cuda), (rn(32) * 0.3).to(cuda)
        m1 = helpers.Decisions.decode(Fh.relu1_masks(vs, gq, w1, b1, B=B, N=N, J=J, groups=G), 0, N, cuda)     # [(B G), N, J, 32]
        pos = gq.double()[None, :, None, :] - vs.double()[:, None, :, :]
        x1 = (torch.sign(pos) * torch.log(pos.abs() + 1)) @ w1.double().T + b1.double()
        bad = x1[(x1 > 0) != m1].abs()
        assert bad.numel() <= 1e-5 * x1.numel() and (bad.numel() == 0 or float(bad.max()) < 1e-6), \
            f"{bad.numel()} layer-1 decisions differ from fp64, worst |pre-activation| {float(bad.max()) if bad.numel() else 0:.2e}"


def test_sampler_gradient_on_cell_boundaries(cuda):
    """LABELLED SANITY CASE: sample positions placed EXACTLY on pixel boundaries, where F.grid_sample's
    position gradient is the slope of one of two cells and rounding decides which.  With the cells the kernel chose
    (smml_bilinear_corners_f32) imposed on the fp64 oracle, values and BOTH gradients agree to 1e-5 - i.e. the kernel's
    gradient is the exact derivative of the branch it took, and nothing else about such inputs needs an exemption."""
    from oracle.deform import bilinear_gather
    B, G, Hh, Ww, cg, J = 2, 8, 16, 16, 16, 64
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(B, Hh, Ww, G * cg, generator=gen)
    k = torch.randint(-1, Ww + 1, (B * G, J, 2), generator=gen).float()
    vs = (2 * k + 1) / Ww - 1                                   # pixel coordinate ((v + 1) W - 1) / 2 = k exactly
    vs[:, ::2] += (torch.rand(B * G, J // 2, 2, generator=gen) - 0.5) * 0.2        # every other sample off the boundary
    w = torch.randn(B, J, G * cg, generator=gen)
    xd, vd = x.to(cuda).requires_grad_(), vs.to(cuda).requires_grad_()
    kv = Fh.bilinear_sample(xd, vd, groups=G, posdim=2)
    (kv * w.to(cuda)).sum().backward()
    cx, cy, _ = Fh.bilinear_corners(vd.detach(), Hh, Ww, 2)
    cells = (cx[:, 0].reshape(B * G, J).long().cpu(), cy[:, 0].reshape(B * G, J).long().cpu())
    on_edge = (vs[:, 1::2] == ((2 * k[:, 1::2] + 1) / Ww - 1)).all()
    assert bool(on_edge)
    xr, vr = x.double().requires_grad_(), vs.double().requires_grad_()
    feats = xr.reshape(B, Hh, Ww, G, cg).permute(0, 3, 1, 2, 4).reshape(B * G, Hh, Ww, cg)
    ref = bilinear_gather(feats, vr[..., 0], vr[..., 1], cells).reshape(B, G, J, cg).permute(0, 2, 1, 3).reshape(B, J, G * cg)
    (ref * w.double()).sum().backward()
    _assert_close("boundary kv", kv, ref, 1e-5)
    _assert_close("boundary dx", xd.grad, xr.grad, 1e-5)
    _assert_close("boundary dvs", vd.grad, vr.grad, 1e-5)


def test_colsum_shapes(cuda):
    """Column sums (bias gradients, landmark / pooler means): the 16-byte-load kernel (C % 4 == 0, 256 % (C / 4) == 0) and the scalar
    one, row counts that are not multiples of the row chunks, several batches, a scale."""
    gen = torch.Generator().manual_seed(8)
    for (nb, R, C) in [(1, 80000, 128), (3, 1000, 512), (2, 77, 128), (1, 5, 4), (2, 300, 36), (1, 10240, 64), (4, 4097, 256), (2, 33, 1024)]:
        x = torch.randn(nb, R, C, generator=gen)
        got = Fh.colsum(x.to(cuda), 0.25)
        ref = x.double().sum(dim=1) * 0.25
        _assert_close(f"colsum {nb}x{R}x{C}", got, ref, 2e-6 * max(1.0, (R / 100) ** 0.5))
