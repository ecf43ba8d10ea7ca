"""GPU parity at the sizes BASELINE.json's configs name (cfg 2, 4, 5; cfg 1 is the CPU plumbing case and cfg 3's shapes are
test_mil_branch_larger_grid_vs_oracle + test_coattention[coattn_L200_S4096]).  The reference itself cannot run these sizes
(N is hard-wired to 2500, DeformCrossTransMIL.py:104); the oracle - pinned to the reference at N = 2500 by the golden
vectors - is the checker, run in fp32 and fp64 on the host for the calibrated tolerance of tests/helpers.py."""
import math

import pytest
import torch

import helpers
import oracle.deform as odeform
import oracle.mil as omil
from helpers import assert_calibrated, assert_close, decision_tap, params_for, rel_err, smml, synth
from oracle.losses import batch_loss, orthogonal_loss
from oracle.mil import deform_pathomic_net
from oracle.nystrom import nystrom_attention
from test_gpu_parity import _compare_param_grads, _load, cpb_probe
from test_oracle_golden import ZERO_GRADS, pathomic_args

pytestmark = pytest.mark.gpu
Fh = smml.functional


def _nystrom_vs_oracle(cuda, tag, B, n, in_dtype=torch.float32, seed=21):
    mod = smml.NystromAttention(dim=512, dim_head=64, heads=8, num_landmarks=256, pinv_iterations=6, residual=True, dropout=0.1)
    params = params_for(mod, seed, tag)
    mod = _load(mod, params, cuda)
    x = (synth.normal((B, n, 512), seed, tag + ":x") * 0.5).to(in_dtype).float()     # values representable in the bag's dtype
    wo = synth.normal((B, n, 512), seed, tag + ":wo")
    run = {}
    for dt in (torch.float32, torch.float64):
        pr = {k: v.clone().to(dt).requires_grad_() for k, v in params.items()}
        xr = x.clone().to(dt).requires_grad_()
        o = nystrom_attention(xr, pr, heads=8, dim_head=64, num_landmarks=256)
        (o * wo.to(dt)).sum().backward()
        run[dt] = (o.detach(), xr.grad, pr)
    xd = x.to(cuda).to(in_dtype).requires_grad_()
    out = mod(xd)
    (out.float() * wo.to(cuda)).sum().backward()
    r32, r64 = run[torch.float32], run[torch.float64]
    if in_dtype == torch.float32:
        assert_calibrated(tag + " out", out, r32[0], r64[0])
        assert_calibrated(tag + " dx", xd.grad, r32[1], r64[1])
        for k, p in mod.named_parameters():
            assert_calibrated(tag + " d" + k, p.grad, r32[2][k].grad, r64[2][k].grad)
    else:
        # a 16-bit bag selects the 16-bit compute mode of the block (operands rounded to bf16 / fp16 on the matrix pipe, fp32
        # accumulation): tolerance per dtype as stated in tests/test_gpu_attn16.py; the gradient comes back in the bag's dtype
        tol = 1e-2 if in_dtype == torch.bfloat16 else 2e-3
        gtol = tol if in_dtype == torch.bfloat16 else 5e-3     # fp16 mode: forward products fp16, gradient products bf16 (tests/test_gpu_attn16.py GRAD_TOL_FP16)
        assert xd.grad.dtype == in_dtype
        assert_close(tag + " out (16-bit mode)", out, r64[0], tol)
        assert_close(tag + " dx (16-bit mode)", xd.grad.float(), r64[1], gtol + (2.0 ** -8 if in_dtype == torch.bfloat16 else 2.0 ** -11))
        for k, p in mod.named_parameters():
            assert_close(tag + " d" + k + " (16-bit mode)", p.grad, r64[2][k].grad, 2 * gtol)


@pytest.mark.parametrize("B,in_dtype", [(1, torch.float32), (2, torch.bfloat16)])
def test_cfg2_nystrom_4096x512_m256(cuda, B, in_dtype):
    """BASELINE config 2: NystromAttention fwd + bwd, bag 4096 x 512, 256 landmarks (l = 16, no padding): the fp32 bag on the
    exact path (1e-4 gate) and the bf16 bag the config names on the 16-bit matrix pipe (two bags: the batch-global max of the
    pseudo-inverse initialisation couples them, NystromAttention.py:26)."""
    _nystrom_vs_oracle(cuda, f"cfg2:{B}:{in_dtype}", B, 4096, in_dtype)


def test_cfg5_nystrom_long_bag_50000(cuda):
    """BASELINE config 5: one bag of 50 000 x 512 instances, 256 landmarks (front padding to 50 176, l = 196), fp16 bag.
    Oracle comparison at the full length (fp32 + fp64 on the host) and size-independent properties: identical bags give
    identical outputs, the front padding is a crop (zeros prepended by the caller give the same rows), linearity in v."""
    n = 50000
    _nystrom_vs_oracle(cuda, "cfg5", 1, n, torch.float16, seed=23)
    mod = smml.NystromAttention(dim=512, dim_head=64, heads=8, num_landmarks=256).to(cuda).eval()
    torch.manual_seed(5)
    x1 = torch.randn(1, n, 512, device=cuda) * 0.5
    with torch.no_grad():
        out2 = mod(torch.cat((x1, x1), 0))
        assert out2.shape == (2, n, 512) and torch.isfinite(out2).all()
        assert torch.equal(out2[0], out2[1])
        out1 = mod(x1)
        pad = (256 - n % 256) % 256
        outp = mod(torch.cat((torch.zeros(1, pad, 512, device=cuda), x1), 1))
        assert_close("front pad = crop", outp[:, -n:], out1, 1e-5)
        # linearity in v: scale the v block of to_qkv by alpha -> (out - bias) scales by alpha
        b = mod.to_out[0].bias
        w = mod.to_qkv.weight.data
        w[2 * 512:] *= 3.0
        out3 = mod(x1)
        w[2 * 512:] /= 3.0
        assert_close("linear in v", out3 - b, 3.0 * (out1 - b), 2e-5)


def _cfg4_once(cuda, B, S, seed, host_fp32):
    """One realisation of BASELINE config 4 (per-rank slice): full two-branch DeformPathomicNet on B bags of S*S x 512 + cross-entropy + both
    BatchLosses + an OrthogonalLoss term, on the HIP path and on the oracle with the HIP path's piecewise-linear decisions imposed:
    fp32 (on the host if host_fp32, else on the GPU's ATen kernels) and fp64.  -> dict of the tensors to compare."""
    args = pathomic_args(input_path_dim=512, batch_size=B)
    net = smml.DeformPathomicNet(args)
    params = params_for(net, seed, "cfg4")
    net = _load(net, params, cuda)
    x_path = synth.bag(B, S * S, 512, seed, "cfg4:bag")
    x_t = synth.normal((B, 59), seed, "cfg4:tumor"); x_i = synth.normal((B, 361), seed, "cfg4:immune")
    label = torch.tensor([2, 0])[:B]

    def total(feats, vt, vi, lg, bl, ol):
        l_t, l_i = bl(lg[3], lg[4]), bl(lg[5], lg[6])
        return (torch.nn.functional.cross_entropy(lg[2], label.to(lg[2].device)) + 0.5 * l_t.sum() + 0.5 * l_i.sum()
                + 0.1 * ol(vt, vi, vi, vt).sum()), l_t, l_i

    # HIP first: the piecewise-linear decisions of its two attention calls (sampler cells, the ReLU decisions of the position bias = the
    # patterns of each pair's linear region) are imposed on the oracle runs, so every gradient is held to the plain rule (helpers.py)
    with decision_tap() as tap:
        feats, vt, vi, lg, _, _, _ = net(x_path=x_path.to(cuda), x_omic=None, x_omic_tumor=x_t.to(cuda), x_omic_immune=x_i.to(cuda))
    bl, ol = smml.BatchLoss(B, 1), smml.OrthogonalLoss()
    loss, l_t, l_i = total(feats, vt, vi, lg, bl, ol)
    loss.backward()
    assert len(tap.decisions()) == 2                       # tumor branch, immune branch (model.py:494,497) - the oracle's order too
    assert lg[4].shape == (B * 8, 2, S // 4, S // 4)
    # the third piecewise-linear place of the model, relu(_fc1(bag)): the decisions of the launch the model itself uses (both branches in
    # one batched launch) are imposed on the oracle as well, and checked against fp64 - they may differ only at rounding-level ties
    ft, fi = net.pathomic_net_tumor._fc1[0], net.pathomic_net_immune._fc1[0]
    with torch.no_grad():
        pf = smml.functional.dual_linear_relu(x_path.to(cuda), ft.weight, ft.bias, fi.weight, fi.bias)
        fc1_masks = [t > 0 for t in pf]
        for m, lin in zip(fc1_masks, (ft, fi)):
            x64 = x_path.to(cuda, torch.float64) @ lin.weight.double().t() + lin.bias.double()
            bad = x64[(x64 > 0) != m].abs()
            assert bad.numel() <= 8 and (bad.numel() == 0 or float(bad.max()) < 2e-6), f"_fc1 ReLU decisions: {bad.numel()} differ from fp64, worst |x| {float(bad.max()) if bad.numel() else 0:.2e}"
    out = {"net": net, "hip": (feats, lg[2], lg[4], lg[6], l_t, l_i, loss), "params": params, "tap": tap, "total": total,
           "inputs": (x_path, x_t, x_i)}
    runs = [("f32", torch.float32, "cpu" if host_fp32 else cuda), ("f64", torch.float64, cuda if S >= 100 else "cpu")]
    if host_fp32:
        runs.insert(1, ("g32", torch.float32, cuda))           # recorded beside the host figure (the back end the reference trains on)
    with cpb_probe() as probe:
        for key, dt, dev in runs:
            p = {k: (v.clone().to(dev, dt).requires_grad_() if v.dtype.is_floating_point else v.to(dev)) for k, v in params.items()}
            odeform.DECISIONS = tap.decisions()
            omil.FC1_DECISIONS = list(fc1_masks)
            o_feats, o_vt, o_vi, o_lg = deform_pathomic_net(x_path.to(dev, dt), x_t.to(dev, dt), x_i.to(dev, dt), p, grid_hw=(S, S), q_chunk=1024 if dev == "cpu" else 2500)
            assert not odeform.DECISIONS and not omil.FC1_DECISIONS
            omil.FC1_DECISIONS = None
            o_loss, o_lt, o_li = total(o_feats, o_vt, o_vi, o_lg, lambda o, v: batch_loss(o, v, B), orthogonal_loss)
            o_loss.backward()
            out[key] = (o_feats.detach().cpu(), o_lg[2].detach().cpu(), o_lg[4].detach().cpu(), o_lg[6].detach().cpu(), o_lt.detach().cpu(),
                        o_li.detach().cpu(), o_loss.detach().cpu(), p)
    out["probe"] = probe
    return out


VALUE_NAMES = ("features", "haz", "vgrid_t", "vgrid_i", "batchloss_t", "batchloss_i", "loss")


def test_cfg4_two_bags_24x24(cuda):
    """Config 4 with two bags on a 24 x 24 grid (the BatchLosses contribute: a 1 x 1 BatchLoss is identically zero): forward values and
    every parameter gradient against the oracle - fp32 on the host (the noise yardstick), fp32 on the GPU's ATen kernels (recorded), fp64."""
    B, S = 2, 24
    r = _cfg4_once(cuda, B, S, 17, host_fp32=True)
    for i, name in enumerate(VALUE_NAMES):
        assert_calibrated("cfg4 " + name, r["hip"][i], r["f32"][i], r["f64"][i], ref32_alt=r["g32"][i])
    net = r["net"]
    with_grad = {k for k, p in net.named_parameters() if p.grad is not None}
    assert with_grad == {k for k, v in r["f64"][7].items() if getattr(v, "grad", None) is not None}, "set of parameters receiving a gradient differs"
    _compare_param_grads(net, r["f32"][7], r["f64"][7], skip=("cls_token",), probe=r["probe"], p32_alt=r["g32"][7])


def test_cfg4_full_fusion_10000x512(cuda):
    """BASELINE config 4 at full size: one bag of 10 000 x 512 (100 x 100 grid, 625 sampled keys) per realisation, FOUR realisations
    (parameters + bag + omic vectors from seeds 17, 23, 29, 31; a realisation costs two full-size oracle runs on the GPU, ~15 s).

    Yardstick (no named tensors, no hand-set bound): one fp32 run is a poor estimate of fp32 noise for gradients that
    are sums over 10 000 queries - over seeds the distance of an fp32 evaluation to fp64 scatters by 5 x for the same tensor
    (profiles/r04_fp32_scatter.txt).  So every tensor is judged on the MEDIAN over the realisations:
        median(HIP error)  <=  max(1e-4, 1.5 x median(error of the oracle evaluated in fp32 on this GPU's ATen kernels)),
    errors = max-norm distance to the fp64 oracle relative to the tensor's scale, same decisions imposed on all three; and no single
    realisation may be further than max(1e-3, 3 x the fp32 oracle's own error on it) (a gross error on one seed cannot hide in a median).  Forward values are asserted per realisation
    under the plain rule.  The data-parallel side of config 4 is tests/test_gpu_data_parallel.py and tests/test_data_parallel_gloo.py."""
    import statistics
    B, S = 1, 100
    seeds = (17, 23, 29, 31)
    e_hip, e_f32 = {}, {}
    first = None
    for seed in seeds:
        r = _cfg4_once(cuda, B, S, seed, host_fp32=False)
        for i, name in enumerate(VALUE_NAMES):
            assert_calibrated(f"cfg4 seed {seed} {name}", r["hip"][i], r["f32"][i], r["f64"][i])
        net, p32, p64 = r["net"], r["f32"][7], r["f64"][7]
        with_grad = {k for k, p in net.named_parameters() if p.grad is not None}
        assert with_grad == {k for k, v in p64.items() if getattr(v, "grad", None) is not None}, "set of parameters receiving a gradient differs"
        for k, p in net.named_parameters():
            if k.endswith("cls_token") or p.grad is None:
                continue
            if k.endswith(ZERO_GRADS):                      # exactly zero in exact arithmetic (softmax shift invariance): held to its natural scale
                helpers.assert_zero_grad(f"seed {seed} d{k}", p.grad, r["probe"].scale(p64[k]))
                continue
            e_hip.setdefault(k, []).append(rel_err(p.grad, p64[k].grad))
            e_f32.setdefault(k, []).append(rel_err(p32[k].grad, p64[k].grad))
        if first is None:
            first = r
        else:
            del r
        torch.cuda.empty_cache()
    failures = []
    for k in e_hip:
        mh, m32 = statistics.median(e_hip[k]), statistics.median(e_f32[k])
        bound = max(1e-4, 1.5 * m32)
        helpers.record("cfg4 d" + k, mh, m32, bound, f"median of {len(seeds)} realisations (noise = fp32 oracle on GPU ATen)")
        if mh > bound:
            failures.append(f"d{k}: median error {mh:.3e} > {bound:.3e} = max(1e-4, 1.5 x fp32 oracle's {m32:.3e}); per seed {['%.2e' % e for e in e_hip[k]]}")
        worst = max(range(len(seeds)), key=lambda i: e_hip[k][i] / max(1e-3, 3.0 * e_f32[k][i]))
        if e_hip[k][worst] > max(1e-3, 3.0 * e_f32[k][worst]):
            failures.append(f"d{k}: error {e_hip[k][worst]:.3e} on seed {seeds[worst]} (fp32 oracle's own: {e_f32[k][worst]:.3e})")
    assert not failures, f"{len(failures)} parameter gradients out of tolerance:\n  " + "\n  ".join(failures)

    # VERDICT r03 "weak" 2 / r04 item 4: the comparisons above impose the kernels' decisions on the oracle.  Against an fp64 oracle run with
    # NOTHING imposed (first realisation): forward VALUES need no imposed decisions (ReLU and the bilinear interpolant are continuous across
    # their kinks) and are asserted at 1e-4; a GRADIENT may differ from that run by what the decisions that differ are worth - measured
    # exactly as the distance between the imposed and the un-imposed fp64 oracle - plus the plain bound (triangle inequality, asserted so
    # that the conditioning of the claim is explicit; that the decisions which differ are rounding-level ties is asserted where the
    # problem is small enough to hold every pre-activation in fp64: tests/test_gpu_regions.py, tests/test_gpu_parity.py).
    r = first
    net, params, total = r["net"], r["params"], r["total"]
    x_path, x_t, x_i = r["inputs"]
    p = {k: (v.clone().to(cuda, torch.float64).requires_grad_() if v.dtype.is_floating_point else v.to(cuda)) for k, v in params.items()}
    odeform.DECISIONS = None
    omil.FC1_DECISIONS = None
    u_feats, u_vt, u_vi, u_lg = deform_pathomic_net(x_path.to(cuda, torch.float64), x_t.to(cuda, torch.float64), x_i.to(cuda, torch.float64), p,
                                                   grid_hw=(S, S), q_chunk=2500)
    u_loss, _, _ = total(u_feats, u_vt, u_vi, u_lg, lambda o, v: batch_loss(o, v, B), orthogonal_loss)
    u_loss.backward()
    hip = r["hip"]
    for name, got, ref in (("features", hip[0], u_feats), ("haz", hip[1], u_lg[2]), ("vgrid_t", hip[2], u_lg[4]), ("vgrid_i", hip[3], u_lg[6]), ("loss", hip[6], u_loss)):
        assert_close("cfg4 nothing imposed: " + name, got, ref.detach(), 1e-4)
    bad = []
    for k, q in net.named_parameters():
        if q.grad is not None and getattr(p[k], "grad", None) is not None and not k.endswith(("cls_token", "rel_pos_bias.mlp.2.bias")):
            worth = rel_err(r["f64"][7][k].grad, p[k].grad.cpu())              # what the differing decisions move this gradient by
            e = rel_err(q.grad, p[k].grad)
            bound = worth + max(1e-4, 1.5 * e_f32[k][0])
            helpers.record("cfg4 nothing imposed: d" + k, e, worth, bound, "vs un-imposed fp64 (noise column = imposed vs un-imposed fp64 oracle)")
            if e > bound:
                bad.append(f"d{k}: {e:.3e} > {bound:.3e}")
    assert not bad, "gradients vs the un-imposed fp64 oracle:\n  " + "\n  ".join(bad)
