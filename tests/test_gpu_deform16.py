"""GPU parity of the 16-bit compute mode of the fused deformable attention (csrc/deform_attn16.hip; BASELINE config 4 names bf16,
config 5 fp16): DeformCrossAttention2D / 1D(compute_dtype='bf16' | 'fp16') and the fused core against the fp64 oracle.

Tolerance of this mode, stated ONCE here (the way tests/test_gpu_attn16.py states the Nystrom block's): every tensor is compared with
the fp64 evaluation in the max norm relative to the tensor's own scale -
    forward values                 bf16 1.5e-2    fp16 4e-3     (8 / 11 operand mantissa bits on q, k, v, P, h1, W2; the stored scores are fp16 in both)
    gradients                      bf16 3e-2      fp16 3e-2     (gradient-range operands - dS, g = h1 . d bias, the dK / dV / dQ products - are bf16 in
                                                                 BOTH modes: fp32's exponent range without a loss scale; measured worst over the fuzz
                                                                 cases 2.3e-2 on dW2 / dW3 of the position-bias MLP, <= 1e-2 elsewhere)
The six parameter gradients of the position-bias MLP are sums over all (query, key) pairs of terms that carry two bf16 roundings (the stored
d score and g = h1 . d bias): at the sizes of the fuzz cases (1e4 .. 1e5 pairs, random signs) that averages little and the max-norm error
reaches 2.3e-2 (bf16) / 4.3e-2 (fp16 mode: same bf16 gradient operands) on dW2 / dW3 - bound MLP_GRAD_TOL_SMALL there; at the headline size
(5e7 pairs per bag and head) the same gradients sit at <= 4e-3 (test_cfg4_full_fusion_16bit, CFG4_GRAD_TOL).
fp32 accumulation everywhere.  The piecewise-linear decisions the kernels took (sampler cells, both ReLU layers of the position-bias MLP)
are exported and imposed on the oracle exactly as in the fp32-grade tests (tests/helpers.py): layer 1 is the SAME fp32-grade device
function in both modes (its decisions may differ from fp64 only at rounding level, 2e-6); layer 2 is a single-term 16-bit product here,
so its decisions may differ where |W2 h1 + b2| is within the 16-bit rounding of the product (asserted below: 2^-7 / 2^-10 of sum |W2||h1|)."""
import math

import pytest
import torch

import helpers
import oracle.deform as odeform
from helpers import assert_close, decision_tap, params_for, rel_err, smml, synth
from oracle.deform import deform_cross_attention_1d, deform_cross_attention_2d
from test_gpu_parity import _core_reference, cpb_probe

pytestmark = pytest.mark.gpu
Fh = smml.functional

FWD_TOL = {"bf16": 1.5e-2, "fp16": 4e-3}
GRAD_TOL = {"bf16": 3e-2, "fp16": 3e-2}
MLP_GRAD_TOL_SMALL = 6e-2      # position-bias MLP parameter gradients of the SMALL fuzz problems (see the docstring)
L2_MARGIN = {"bf16": 2.0 ** -7, "fp16": 2.0 ** -10}     # layer-2 decisions: |pre-activation| of a flipped unit <= margin x sum |W2| |h1| (+ |b2|)


@pytest.mark.parametrize("tabfwd", [False, "table", "recompute"])
@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_fused_core16_random_shapes(cuda, mode, tabfwd, monkeypatch):
    """The 16-bit fused core (forward + the three backward passes) on random ragged shapes - N, J off the 32 / 128 tiles, one or two
    heads per offset group, 1-D and 2-D positions, with and without dropout - against plain torch in fp64 with the kernels' own
    ReLU decisions and dropout mask imposed.  tabfwd: cpb_table='forward' with the layer-2 decisions of the backward from the mask table / recomputed."""
    if tabfwd:
        monkeypatch.setattr(Fh, "TABLE_FORWARD_MASKS", tabfwd)
    import os
    gen = torch.Generator().manual_seed(int(os.environ.get("SMML_FUZZ_SEED", "4321")))      # soak: SMML_FUZZ_CASES=60 SMML_FUZZ_SEED=7 pytest -k core16_random
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen))
    forced = [(1, 1, 65, 8, 1, 0.0), (2, 33, 5, 4, 2, 0.25), (1, 129, 33, 8, 2, 0.0), (1, 300, 1, 8, 2, 0.0)]
    nrand = int(os.environ.get("SMML_FUZZ_CASES", "8"))
    worst = {}
    for case in range(nrand + len(forced)):
        B, N, J = ri(1, 3), ri(1, 300), ri(2, 90)
        groups = (4, 8)[ri(0, 1)]
        heads, PD, p_drop = 8, ri(1, 2), (0.0, 0.25)[ri(0, 1)]
        if case >= nrand:
            B, N, J, groups, PD, p_drop = forced[case - nrand]
        rn = lambda *s: torch.randn(*s, generator=gen)
        t = dict(q=rn(B, N, 512) * 0.4, k=rn(B, J, 512) * 0.4, v=rn(B, J, 512), vs=torch.rand(B * groups, J, PD, generator=gen) * 2.4 - 1.2,
                 gq=torch.rand(N, PD, generator=gen) * 2 - 1, w1=rn(32, PD) * 0.7, b1=rn(32) * 0.3, w2=rn(32, 32) * 0.25, b2=rn(32) * 0.2,
                 w3=rn(heads // groups, 32) * 0.3, b3=rn(heads // groups) * 0.1)
        wo = rn(B, N, 512)
        names = ("q", "k", "v", "vs", "gq", "w1", "b1", "w2", "b2", "w3", "b3")
        dev = {n: x.to(cuda).requires_grad_() for n, x in t.items()}
        seed = 170 + case
        smml.functional.DECISION_TAP = tapped = []
        try:
            # tabfwd: the forward's bias comes from the table (cpb_table='forward'), the backward recomputes layer 2 (always one bf16 term) and
            # exports the decisions it took - everything below then applies unchanged
            tkw = dict(cpb_table="forward", cpb_table_pmax=Fh.table_pmax(1.0, 1.2)) if tabfwd else {}
            # (cpb_regions=False: this test pins the per-pair MLP kernels of the 16-bit core; the region form of the same core is covered by
            # tests/test_gpu_regions.py and by the module-level tests below, which take it wherever it applies)
            out = Fh.deform_attention(*(dev[n] for n in names), heads=heads, groups=groups, scale=0.125, dropout_p=p_drop,
                                      dropout_seed=seed, compute_dtype=mode, cpb_regions=False, **tkw)
        finally:
            smml.functional.DECISION_TAP = None
        (out * wo.to(cuda)).sum().backward()
        keep = Fh.deform_attention_dropout_mask(B, N, J, heads, p_drop, seed, cuda) if p_drop else None
        a = tapped[0]
        m1 = helpers.Decisions.decode(Fh.relu1_masks(a["vs"], a["gq"], a["w1"], a["b1"], B=B, N=N, J=J, groups=groups), 0, N, cuda)
        m2 = helpers.Decisions.decode(Fh.relu_masks_rows(a["masks2"])[:, ::heads // groups].reshape(B * groups, J, 2, -1), 0, N, cuda)
        r = {n: x.to(cuda, torch.float64).requires_grad_() for n, x in t.items()}
        o = _core_reference(*(r[n] for n in names), heads, groups, 0.125, keep, 1.0 / (1.0 - p_drop), masks=(m1, m2))
        (o * wo.to(cuda, torch.float64)).sum().backward()
        with torch.no_grad():       # the decisions against the exact pre-activations
            pos = r["gq"][None, :, None, :] - r["vs"].view(B * groups, 1, J, PD)
            x1 = (torch.sign(pos) * torch.log(pos.abs() + 1)) @ r["w1"].T + r["b1"]
            h1 = torch.relu(x1)
            x2 = h1 @ r["w2"].T + r["b2"]
            bad1 = x1[(x1 > 0) != m1].abs()
            assert bad1.numel() == 0 or float(bad1.max()) < 2e-6, f"case {case}: layer-1 decision off by {float(bad1.max()):.2e}"
            mag = h1 @ r["w2"].abs().T + r["b2"].abs()                      # sum |W2| |h1| + |b2| per unit
            flip = (x2 > 0) != m2
            if flip.any():
                ratio = float((x2.abs() / mag.clamp_min(1e-30))[flip].max())
                worst["l2 flip / magnitude"] = max(worst.get("l2 flip / magnitude", 0.0), ratio)
                assert ratio < L2_MARGIN["bf16" if tabfwd else mode], f"case {case}: a layer-2 decision differs from fp64 at {ratio:.2e} of the unit's magnitude"
        tag = f"{mode} case {case}: B={B} N={N} J={J} G={groups} PD={PD} p={p_drop}"
        e = rel_err(out, o); worst["out"] = max(worst.get("out", 0.0), e)
        assert_close(tag + " out", out, o, FWD_TOL[mode])
        for n in t:
            if n in ("gq", "b3"):
                continue
            g, g64 = dev[n].grad, r[n].grad
            if float(g64.abs().max()) < 1e-9:               # identically zero in exact arithmetic (one key: dS = 0)
                assert float(g.abs().max()) < 5e-2, f"{tag} d{n}: expected ~0, got {float(g.abs().max()):.3e}"
                continue
            e = rel_err(g, g64); worst["d" + n] = max(worst.get("d" + n, 0.0), e)
            assert_close(tag + " d" + n, g, g64, MLP_GRAD_TOL_SMALL if n in ("w1", "b1", "w2", "b2", "w3") else GRAD_TOL[mode])
    print(f"\n[deform16 {mode}{' table forward, masks: ' + tabfwd if tabfwd else ''}] worst relative errors over the fuzz cases: " + ", ".join(f"{k} {v:.2e}" for k, v in worst.items()))


@pytest.mark.parametrize("tabfwd", [False, "table", "recompute"])
@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("train", [False, True])
def test_deform2d_16bit_vs_oracle(cuda, mode, train, tabfwd, monkeypatch):
    """DeformCrossAttention2D(compute_dtype=...) on a 20 x 20 grid, eval and train (dropout 0.1 with the kernel's exported mask): output,
    vgrid (untouched by the mode: exact), input and parameter gradients against the fp64 oracle with the kernels' decisions imposed."""
    if tabfwd:
        monkeypatch.setattr(Fh, "TABLE_FORWARD_MASKS", tabfwd)
    B, Hh, Ww, C = 2, 20, 20, 128
    N = Hh * Ww
    tag = f"d2d16:{mode}:{int(train)}"
    mod = smml.DeformCrossAttention2D(dim=C, dropout=0.1, grid_hw=(Hh, Ww), compute_dtype=mode, cpb_table="forward" if tabfwd else False)
    params = params_for(mod, 31, tag)
    mod.load_state_dict(params)
    mod = mod.to(cuda).train(train)
    x1 = synth.normal((B, C, N), 31, tag + ":x1"); x2 = synth.normal((B, C, N), 31, tag + ":x2")
    w_out = synth.normal((B, C, N), 31, tag + ":wo")
    torch.manual_seed(99)
    ad, bd = x1.to(cuda).requires_grad_(), x2.to(cuda).requires_grad_()
    with decision_tap() as tap:
        o, vg = mod(ad, bd, return_vgrid=True)
    (o * w_out.to(cuda)).sum().backward()
    J = vg.shape[-1] * vg.shape[-2]
    keep = Fh.deform_attention_dropout_mask(B, N, J, 8, 0.1, mod.last_dropout_seed, cuda).cpu() if train else None
    dt = torch.float64
    pref = {k: v.clone().to(dt).requires_grad_() for k, v in params.items()}
    a, b = x1.clone().to(dt).requires_grad_(), x2.clone().to(dt).requires_grad_()
    odeform.DECISIONS = tap.decisions()
    kw = dict(attn_keep=keep, dropout_p=0.1) if train else {}
    o_ref, vg_ref = deform_cross_attention_2d(a, b, pref, grid_hw=(Hh, Ww), **kw)
    (o_ref * w_out.to(dt)).sum().backward()
    assert_close(tag + " vgrid", vg, vg_ref, 1e-5)
    assert_close(tag + " out", o, o_ref, FWD_TOL[mode])
    assert_close(tag + " dx1", ad.grad, a.grad, GRAD_TOL[mode])
    assert_close(tag + " dx2", bd.grad, b.grad, GRAD_TOL[mode])
    for k, p in mod.named_parameters():
        if k.endswith("rel_pos_bias.mlp.2.bias"):           # zero in exact arithmetic (softmax shift invariance)
            continue
        assert_close(tag + " d" + k, p.grad, pref[k].grad, MLP_GRAD_TOL_SMALL if "rel_pos_bias" in k else GRAD_TOL[mode])   # 2e5 pairs: a small problem
    # the mode changes nothing outside the fused core: same vgrid bits as the fp32-grade module
    ref_mod = smml.DeformCrossAttention2D(dim=C, dropout=0.1, grid_hw=(Hh, Ww))
    ref_mod.load_state_dict(params)
    ref_mod = ref_mod.to(cuda).eval()
    with torch.no_grad():
        o32, vg32 = ref_mod(ad.detach(), bd.detach(), return_vgrid=True)
    assert torch.equal(vg32, vg.detach())
    if not train:
        assert_close(tag + " out vs fp32-grade path", o, o32, FWD_TOL[mode])


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_deform1d_16bit_vs_oracle(cuda, mode):
    """DeformCrossAttention1D(compute_dtype=...): two heads per offset group, 1-D positions, n = 129 (front of a ragged tile)."""
    B, n, C = 2, 129, 128
    tag = f"d1d16:{mode}"
    mod = smml.DeformCrossAttention1D(dim=C, downsample_factor=4, offset_scale=2, offset_kernel_size=6, compute_dtype=mode)
    params = params_for(mod, 33, tag)
    mod.load_state_dict(params)
    mod = mod.to(cuda).eval()
    x1 = synth.normal((B, C, n), 33, tag + ":x1"); x2 = synth.normal((B, C, n), 33, tag + ":x2")
    w_out = synth.normal((B, C, n), 33, tag + ":wo")
    ad, bd = x1.to(cuda).requires_grad_(), x2.to(cuda).requires_grad_()
    with decision_tap() as tap:
        o = mod(ad, bd)
    (o * w_out.to(cuda)).sum().backward()
    dt = torch.float64
    pref = {k: v.clone().to(dt).requires_grad_() for k, v in params.items()}
    a, b = x1.clone().to(dt).requires_grad_(), x2.clone().to(dt).requires_grad_()
    odeform.DECISIONS = tap.decisions()
    o_ref, _ = deform_cross_attention_1d(a, b, pref, downsample_factor=4, offset_scale=2, offset_kernel_size=6)
    (o_ref * w_out.to(dt)).sum().backward()
    assert_close(tag + " out", o, o_ref, FWD_TOL[mode])
    assert_close(tag + " dx1", ad.grad, a.grad, GRAD_TOL[mode])
    assert_close(tag + " dx2", bd.grad, b.grad, GRAD_TOL[mode])
    for k, p in mod.named_parameters():
        if k.endswith("rel_pos_bias.mlp.2.bias") or pref[k].grad is None:
            continue
        assert_close(tag + " d" + k, p.grad, pref[k].grad, MLP_GRAD_TOL_SMALL if "rel_pos_bias" in k else GRAD_TOL[mode])


def test_core16_gradients_are_run_to_run_identical(cuda):
    """Fixed-order reductions in the 16-bit mode too: two runs give the same bits (dropout on)."""
    gen = torch.Generator().manual_seed(22)
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
        out = Fh.deform_attention(*(dev[n] for n in names), heads=heads, groups=groups, scale=0.125, dropout_p=0.1, dropout_seed=5,
                                  compute_dtype="bf16")
        (out * wo).sum().backward()
        torch.cuda.synchronize()
        runs.append({n: dev[n].grad.clone() for n in names if n != "gq"} | {"out": out.detach().clone()})
    for n in runs[0]:
        assert torch.equal(runs[0][n], runs[1][n]), f"{n} differs between two runs"


def test_core16_dropout_decisions_ride_in_the_saved_scores(cuda):
    """Train mode: the keep decisions the backward reads from the lowest bit of the saved 16-bit scores are the exported mask's -
    checked through the gradient: with v = const every dropped pair contributes nothing to dv, so dv[key] = sum_q keep P dO exactly as
    the exported mask says (fp64 recomputation from the kernel's own saved probabilities is not needed: compare with the oracle core)."""
    gen = torch.Generator().manual_seed(5)
    B, N, J, heads, groups, PD, p = 1, 257, 70, 8, 8, 2, 0.3
    rn = lambda *s: torch.randn(*s, generator=gen)
    t = dict(q=rn(B, N, 512) * 0.3, k=rn(B, J, 512) * 0.3, v=rn(B, J, 512), vs=torch.rand(B * groups, J, PD, generator=gen) * 2 - 1,
             gq=torch.rand(N, PD, generator=gen) * 2 - 1, w1=rn(32, PD) * 0.5, b1=rn(32) * 0.3, w2=rn(32, 32) * 0.2, b2=rn(32) * 0.2,
             w3=rn(1, 32) * 0.3, b3=rn(1) * 0.1)
    names = ("q", "k", "v", "vs", "gq", "w1", "b1", "w2", "b2", "w3", "b3")
    for mode in ("bf16", "fp16"):
        dev = {n: x.to(cuda).requires_grad_() for n, x in t.items()}
        out = Fh.deform_attention(*(dev[n] for n in names), heads=heads, groups=groups, scale=0.125, dropout_p=p, dropout_seed=77, compute_dtype=mode)
        out.sum().backward()
        keep = Fh.deform_attention_dropout_mask(B, N, J, heads, p, 77, cuda)
        r = {n: x.to(cuda, torch.float64).requires_grad_() for n, x in t.items()}
        o = _core_reference(*(r[n] for n in names), heads, groups, 0.125, keep, 1.0 / (1.0 - p))
        o.sum().backward()
        # a wrong keep bit anywhere moves dv by O(P) of single pairs: far above the mode's rounding
        assert_close(f"{mode} dv under dropout", dev["v"].grad, r["v"].grad, GRAD_TOL[mode])
        assert_close(f"{mode} out under dropout", out, o, FWD_TOL[mode] * 2)


# Full-model bounds of the mode (BASELINE config 4 as stated: bf16 compute of the deformable path - the fused core and its output projection;
# everything upstream of the sample positions stays fp32-grade, so the sampler's integer path is bit-identical to the fp32-grade model's).  Forward values and the losses: FWD_TOL.  Every parameter gradient: CFG4_GRAD_TOL of its own scale - measured at the
# full size (1 x 100 x 100, bf16) 7.0e-3 on tumor to_offsets.2.weight, <= 4e-3 on all others; fp16 on the 24 x 24 grid <= 6e-4.
CFG4_GRAD_TOL = {"bf16": 2e-2, "fp16": 1e-2}


@pytest.mark.parametrize("B,S,mode,table", [(2, 24, "bf16", False), (1, 100, "bf16", False), (2, 24, "fp16", False),
                                              (2, 24, "bf16", "forward"), (1, 100, "bf16", "forward")])
def test_cfg4_full_fusion_16bit(cuda, B, S, mode, table):
    """BASELINE config 4 as it is stated - full two-branch DeformPathomicNet on bags of 10 000 x 512 with the deformable attention computing
    in bf16 (args.deform_compute_dtype; fp32 master parameters, fp32 inputs / outputs / gradients) + cross-entropy + both BatchLosses + an
    OrthogonalLoss term: forward and every parameter gradient against the fp64 oracle with the kernels' decisions imposed (one oracle
    run, on the GPU's fp64 ATen kernels at the full size).  The two-bag case (BatchLosses contribute) runs on a 24 x 24 grid."""
    from oracle.losses import batch_loss, orthogonal_loss
    from oracle.mil import deform_pathomic_net
    from test_oracle_golden import ZERO_GRADS, pathomic_args
    # table = 'forward' (args.deform_cpb_table): the forward's position bias from the table, the backward's layer-2 decisions from the mask table
    args = pathomic_args(input_path_dim=512, batch_size=B, deform_compute_dtype=mode, deform_cpb_table=table)
    net = smml.DeformPathomicNet(args)
    params = params_for(net, 17, "cfg4")
    net.load_state_dict(params)
    net = net.to(cuda).eval()
    assert net.pathomic_net_tumor.layer3.attn2d.compute_dtype == mode
    x_path = synth.bag(B, S * S, 512, 17, "cfg4:bag")
    x_t = synth.normal((B, 59), 17, "cfg4:tumor"); x_i = synth.normal((B, 361), 17, "cfg4:immune")
    label = torch.tensor([2, 0])[:B]

    def total(feats, vt, vi, lg, bl, ol):
        l_t, l_i = bl(lg[3], lg[4]), bl(lg[5], lg[6])
        return (torch.nn.functional.cross_entropy(lg[2], label.to(lg[2].device)) + 0.5 * l_t.sum() + 0.5 * l_i.sum()
                + 0.1 * ol(vt, vi, vi, vt).sum()), l_t, l_i

    with decision_tap() as tap:
        feats, vt, vi, lg, _, _, _ = net(x_path=x_path.to(cuda), x_omic=None, x_omic_tumor=x_t.to(cuda), x_omic_immune=x_i.to(cuda))
    loss, l_t, l_i = total(feats, vt, vi, lg, smml.BatchLoss(B, 1), smml.OrthogonalLoss())
    loss.backward()
    dev = cuda if S >= 100 else "cpu"
    dt = torch.float64
    p = {k: (v.clone().to(dev, dt).requires_grad_() if v.dtype.is_floating_point else v.to(dev)) for k, v in params.items()}
    odeform.DECISIONS = tap.decisions()
    o_feats, o_vt, o_vi, o_lg = deform_pathomic_net(x_path.to(dev, dt), x_t.to(dev, dt), x_i.to(dev, dt), p, grid_hw=(S, S), q_chunk=1024)
    assert not odeform.DECISIONS
    o_loss, o_lt, o_li = total(o_feats, o_vt, o_vi, o_lg, lambda o, v: batch_loss(o, v, B), orthogonal_loss)
    o_loss.backward()
    tag = f"cfg4/{mode}{'/table-forward' if table else ''} {B}x{S}x{S}"
    for name, got, ref in (("features", feats, o_feats), ("haz", lg[2], o_lg[2]), ("vgrid_t", lg[4], o_lg[4]), ("vgrid_i", lg[6], o_lg[6]),
                           ("loss", loss, o_loss)):
        assert_close(f"{tag} {name}", got, ref, FWD_TOL[mode])
    if B > 1:
        assert_close(f"{tag} batchloss_t", l_t, o_lt, FWD_TOL[mode]); assert_close(f"{tag} batchloss_i", l_i, o_li, FWD_TOL[mode])
    with_grad = {k for k, q in net.named_parameters() if q.grad is not None}
    assert with_grad == {k for k, v in p.items() if getattr(v, "grad", None) is not None}, "set of parameters receiving a gradient differs"
    bad = []
    for k, q in net.named_parameters():
        if q.grad is None or k.endswith(ZERO_GRADS) or k.endswith("cls_token"):
            continue
        tol = CFG4_GRAD_TOL[mode]
        e = rel_err(q.grad, p[k].grad)
        helpers.record(f"{tag} d{k}", e, None, tol, "max vs fp64")
        if not e <= tol:
            bad.append(f"{k}: {e:.2e} > {tol:.1e}")
    assert not bad, f"{len(bad)} parameter gradients out of tolerance:\n  " + "\n  ".join(bad)
