"""GPU tests of the per-linear-region evaluation of the position bias (csrc/cpb_regions.h; include/smml.h "region" entry points).

The position-bias MLP of models/DeformableAttention2D.py:129-152 is piecewise affine in the signed-log offsets.  These tests pin
  * the TABLES: for any point of the tabulated square the lookup (cell -> region | kink record | sub-cell | "evaluate the MLP") names
    the linear piece whose 64 ReLU decisions are those of an fp64 evaluation of the reference's formula, except where a
    pre-activation is within fp32 rounding of zero; and the piece's (a, c) reproduce the MLP's value;
  * the KERNELS: forward and backward through the regions against the per-pair MLP kernels (the round-1..4 path, itself pinned to the
    reference's goldens) on the same inputs, and against plain torch in fp64 with the kernels' decisions imposed;
  * determinism and the capacity fallbacks.
No new tolerance: the bounds are those of tests/test_gpu_parity.py."""
import pytest
import torch

import helpers
from helpers import assert_calibrated, smml, synth
from test_gpu_parity import _core_reference

pytestmark = pytest.mark.gpu
Fh = smml.functional
NAMES = ("q", "k", "v", "vs", "gq", "w1", "b1", "w2", "b2", "w3", "b3")


def _mlp_weights(kind, gen):
    rn = lambda *s: torch.randn(*s, generator=gen)
    if kind == "bench":        # bench.py's parameters of the 2-D module's position bias
        shapes = {"mlp.0.0.weight": (32, 2), "mlp.0.0.bias": (32,), "mlp.1.0.weight": (32, 32), "mlp.1.0.bias": (32,), "mlp.2.weight": (1, 32),
                  "mlp.2.bias": (1,)}
        p = synth.fill_params({"layer3.attn2d.rel_pos_bias." + k: v for k, v in shapes.items()}, 42, "bench")
        return [p["layer3.attn2d.rel_pos_bias." + k] for k in shapes]
    if kind == "torch":        # nn.Linear's default initialisation (what a fresh reference module holds)
        torch.manual_seed(3)
        l1, l2, l3 = torch.nn.Linear(2, 32), torch.nn.Linear(32, 32), torch.nn.Linear(32, 1)
        return [t.detach().clone() for t in (l1.weight, l1.bias, l2.weight, l2.bias, l3.weight, l3.bias)]
    if kind == "star":         # every layer-1 kink through (almost) one point: many cells crossed by several kinks
        return [rn(32, 2) * 0.7, rn(32) * 1e-3, rn(32, 32) * 0.25, rn(32) * 0.2, rn(1, 32) * 0.3, rn(1) * 0.1]
    return [rn(32, 2) * 0.7, rn(32) * 0.3, rn(32, 32) * 0.25, rn(32) * 0.2, rn(1, 32) * 0.3, rn(1) * 0.1]


def _lookup(view, p):
    """The region kernels' lookup restated in torch: p [n, 2] fp32 on the device -> region id per point (0xFFFF: evaluate the MLP) and the
    kind of the point's first-level cell (0 region, 1 kink record, 2 refined, 3 MLP)."""
    G, SUB, S0, E0, NE = Fh.REGION_GRID, Fh.REGION_SUB, Fh.REGION_CODE_SUB0, Fh.REGION_CODE_EDGE0, Fh.REGION_EDGES
    cs, co = view["cs"], view["co"]
    u = torch.addcmul(torch.full_like(p, co), p, torch.full_like(p, cs))
    cell = u.to(torch.int32).clamp(0, G - 1).long()
    e = view["t0"][cell[:, 1], cell[:, 0]].long() & 0xFFFF
    kind0 = torch.where(e < S0, 0, torch.where(e < E0, 2, torch.where(e < E0 + NE, 1, 3)))
    is_sub = (e >= S0) & (e < E0)
    if bool(is_sub.any()):
        sx = ((u - cell.float()) * SUB).to(torch.int32).clamp(0, SUB - 1).long()
        e1 = view["t1"][e[is_sub] - S0, sx[is_sub, 1] * SUB + sx[is_sub, 0]].long() & 0xFFFF
        e = e.clone()
        e[is_sub] = e1
    rid = e.clone()
    is_edge = (e >= E0) & (e < E0 + NE)
    if bool(is_edge.any()):
        rec = view["edge"][e[is_edge] - E0]
        pe = p[is_edge]
        g = torch.addcmul(torch.addcmul(rec[:, 2], rec[:, 1], pe[:, 1]), rec[:, 0], pe[:, 0])
        bits = rec[:, 3].contiguous().view(torch.int32).long() & 0xFFFFFFFF
        rid[is_edge] = torch.where(g > 0, bits >> 16, bits & 0xFFFF)
    return rid, kind0


@pytest.mark.parametrize("kind", ["bench", "torch", "random", "star"])
def test_region_tables_name_the_right_linear_piece(cuda, kind):
    gen = torch.Generator().manual_seed(11)
    w = [t.to(cuda).contiguous() for t in _mlp_weights(kind, gen)]
    pmax = 1.25
    tables = Fh.cpb_regions_build(*w, pmax)
    torch.cuda.synchronize()
    view = Fh.region_tables_view(tables)
    assert view["overflow"] == 0, f"capacity overflow flags {view['overflow']:#x}"
    assert 1 <= view["n_regions"] <= Fh.REGION_RCAP
    print(f"[{kind}] regions {view['n_regions']} records {view['n_edge']} refined cells {view['n_sub']} line candidates {view['n_cand']}")
    n = 2_000_000
    p = ((torch.rand(n, 2, generator=gen) * 2 - 1) * (pmax * 0.995)).to(cuda)
    rid, kindv = _lookup(view, p)
    none = rid == 0xFFFF
    assert float(none.float().mean()) < 2e-3, f"{float(none.float().mean()):.2e} of the points fall to the MLP path"
    w64 = [t.double() for t in w]
    p64 = p.double()
    x1 = p64 @ w64[0].T + w64[1]
    x2 = torch.relu(x1) @ w64[2].T + w64[3]
    val = torch.relu(x2) @ w64[4].T + w64[5]
    pat = view["pat"][rid.clamp_max(view["n_regions"] - 1)]
    shifts = torch.arange(32, device=cuda)
    d1 = ((pat[:, None] >> shifts) & 1).bool()
    d2 = ((pat[:, None] >> (shifts + 32)) & 1).bool()
    ok = ~none
    for name, x, d in (("layer 1", x1, d1), ("layer 2", x2, d2)):
        bad = (((x > 0) != d) & ok[:, None])
        worst = float(x[bad].abs().max()) if bool(bad.any()) else 0.0
        assert worst < 2e-6, f"[{kind}] {name}: a tabulated decision differs from fp64 at |pre-activation| {worst:.2e} ({int(bad.sum())} differ)"
    reg = view["reg"][rid.clamp_max(view["n_regions"] - 1)].double()
    got = reg[:, 0] * p64[:, 0] + reg[:, 1] * p64[:, 1] + reg[:, 2]
    err = float(((got - val[:, 0]).abs() * ok).max()) / max(float(val.abs().max()), 1e-30)
    assert err < 2e-6, f"[{kind}] a region's (a, c) is {err:.2e} off the MLP's value"
    share = [float((kindv == k).float().mean()) for k in range(4)]
    print(f"[{kind}] share of points: region {share[0]:.4f} kink record {share[1]:.4f} refined {share[2]:.4f} MLP {share[3]:.2e}; "
          f"after refinement MLP {float(none.float().mean()):.2e}")


def _problem(gen, B, N, J, heads, wkind="random", vs_scale=1.2):
    rn = lambda *s: torch.randn(*s, generator=gen)
    w = _mlp_weights(wkind, gen)
    return dict(q=rn(B, N, heads * 64) * 0.4, k=rn(B, J, heads * 64) * 0.4, v=rn(B, J, heads * 64),
                vs=torch.rand(B * heads, J, 2, generator=gen) * (2 * vs_scale) - vs_scale, gq=torch.rand(N, 2, generator=gen) * 2 - 1,
                w1=w[0], b1=w[1], w2=w[2], b2=w[3], w3=w[4], b3=w[5])


def _run(t, cuda, wo, regions, p_drop=0.0, seed=3, heads=8, tap=False, compute_dtype=None):
    dev = {n: x.to(cuda).requires_grad_(n != "gq") for n, x in t.items()}
    if tap:
        Fh.DECISION_TAP = tapped = []
    try:
        out = Fh.deform_attention(*(dev[n] for n in NAMES), heads=heads, groups=heads, scale=0.125, dropout_p=p_drop, dropout_seed=seed,
                                  cpb_regions=regions, compute_dtype=compute_dtype)
    finally:
        if tap:
            Fh.DECISION_TAP = None
    (out * wo).sum().backward()
    torch.cuda.synchronize()
    res = {n: dev[n].grad.detach().clone() for n in NAMES if n != "gq"} | {"out": out.detach().clone()}
    return (res, tapped) if tap else res


@pytest.mark.parametrize("wkind,B,N,J,p_drop", [("random", 2, 700, 150, 0.1), ("bench", 1, 2500, 144, 0.0), ("star", 1, 333, 70, 0.0),
                                                ("torch", 3, 129, 33, 0.25), ("random", 1, 1, 65, 0.0), ("random", 2, 5, 1, 0.0),
                                                ("random", 1, 200, 900, 0.1), ("bench", 1, 130, 1601, 0.0)])      # > 768 keys: key groups in the bias backward
def test_region_core_matches_per_pair_mlp(cuda, wkind, B, N, J, p_drop):
    """Same inputs through the region kernels and through the per-pair MLP kernels: every output agrees to fp32 rounding (both sit
    within the parity gate of the fp64 oracle; here they are held against EACH OTHER at 2e-5 of each tensor's scale - an order
    below the gate - except where the two paths took different ReLU decisions, which the decision-imposed test below covers)."""
    gen = torch.Generator().manual_seed(100 + N + J)
    t = _problem(gen, B, N, J, 8, wkind)
    wo = torch.randn(B, N, 512, generator=gen).to(cuda)
    a = _run(t, cuda, wo, True, p_drop)
    b = _run(t, cuda, wo, False, p_drop)
    for n in a:
        scale = max(float(b[n].abs().max()), 1e-30)
        err = float((a[n] - b[n]).abs().max()) / scale
        if n == "b3" or float(b[n].abs().max()) < 1e-9:
            continue
        # parameter gradients of the bias MLP are sums over all pairs in which single ReLU flips at rounding level show (test_gpu_parity)
        tol = 2e-5 if n in ("out", "q", "k", "v") else 5e-4
        if J > 768 and n not in ("out", "q", "k", "v"):   # few queries per key: ONE pair decided differently at a pre-activation within rounding
            tol = 2e-2 if n == "vs" else 2e-3             # of zero shows at 1 / N of a key's d vs (these shapes are also in the decision-imposed
                                                          # fp64 test below, which is the gate)
        assert err <= tol, f"{wkind} {B}x{N}x{J}: {n} differs by {err:.2e} of its scale between the region and the per-pair kernels"
    a2 = _run(t, cuda, wo, True, p_drop)
    for n in a:
        assert torch.equal(a[n], a2[n]), f"{n}: the region path is not run-to-run identical"
    # Parameter sets with more than 2048 linear regions (none of the initialisations tried has more than ~1 850) keep the regions beyond
    # the 2048th in global memory: coefficients read there, moments added there.  SmmlDeformOpts.region_lds_cap lowers the limit so
    # that ordinary parameters exercise that path: same forward bits, same gradients up to the order of the moment sums.
    Fh.REGION_LDS_CAP = 96
    try:
        c = _run(t, cuda, wo, True, p_drop)
    finally:
        Fh.REGION_LDS_CAP = 0
    for n in a:
        scale = max(float(a[n].abs().max()), 1e-30)
        err = float((a[n] - c[n]).abs().max()) / scale
        tol = 0.0 if n in ("out",) else (2e-6 if n in ("q", "k", "v") else 2e-5)
        assert err <= tol, f"{wkind} {B}x{N}x{J}: {n} differs by {err:.2e} between LDS-resident and global-memory regions"


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("wkind,B,N,J,p_drop", [("random", 2, 700, 150, 0.1), ("bench", 1, 1500, 144, 0.0), ("torch", 3, 129, 33, 0.25), ("random", 2, 5, 3, 0.0)])
def test_region_core_in_the_16bit_modes(cuda, mode, wkind, B, N, J, p_drop):
    """The region form of the 16-bit compute modes (smml_deform_attn16_region_fwd / _bwd: single-term operands, fp16 scores, bf16 d scores,
    the position bias by the SAME fp32 lookup) against the fp32-grade region kernels on the same inputs: same region ids (the lookup does
    not depend on the mode), outputs and gradients within the 16-bit bounds of tests/test_gpu_deform16.py, and run-to-run identical.
    Its parity with the fp64 oracle under imposed decisions is checked at module level (test_deform2d_16bit_vs_oracle,
    test_cfg4_full_fusion_16bit: both take this path)."""
    gen = torch.Generator().manual_seed(200 + N + J)
    t = _problem(gen, B, N, J, 8, wkind)
    wo = torch.randn(B, N, 512, generator=gen).to(cuda)
    a, tap_a = _run(t, cuda, wo, True, p_drop, tap=True)
    b, tap_b = _run(t, cuda, wo, True, p_drop, tap=True, compute_dtype=mode)
    assert tap_b[0].get("region_ids") is not None, "the 16-bit call did not take the region path"
    assert torch.equal(tap_a[0]["region_ids"], tap_b[0]["region_ids"]), "the region ids depend on the compute mode"
    for n in a:
        scale = max(float(a[n].abs().max()), 1e-30)
        if n == "b3" or float(a[n].abs().max()) < 1e-9:       # (d b3 = sum of all d scores: zero in exact arithmetic, rounding in both paths)
            continue
        err = float((a[n] - b[n]).abs().max()) / scale
        tol = 1.5e-2 if n == "out" else (3e-2 if n in ("q", "k", "v", "vs") else 6e-2)
        assert err <= tol, f"{mode} {wkind} {B}x{N}x{J}: {n} differs by {err:.2e} of its scale from the fp32-grade region kernels"
    b2 = _run(t, cuda, wo, True, p_drop, compute_dtype=mode)
    for n in b:
        assert torch.equal(b[n], b2[n]), f"{n}: the 16-bit region path is not run-to-run identical"


def test_region_core_vs_fp64_with_imposed_decisions(cuda):
    """The region kernels against plain torch in fp64 and fp32 with the decisions the kernels stand for (the ReLU patterns of each
    pair's region) imposed on both - the rule of tests/test_gpu_parity.py::test_fused_core_random_shapes, same bounds."""
    gen = torch.Generator().manual_seed(77)
    for case, (wkind, B, N, J, p_drop) in enumerate([("random", 2, 300, 90, 0.0), ("bench", 1, 500, 144, 0.25), ("star", 2, 129, 40, 0.0),
                                                            ("random", 1, 200, 900, 0.1), ("bench", 1, 130, 1601, 0.0)]):
        t = _problem(gen, B, N, J, 8, wkind)
        wo = torch.randn(B, N, 512, generator=gen)
        res, tapped = _run(t, cuda, wo.to(cuda), True, p_drop, seed=17 + case, tap=True)
        m1, m2 = helpers.decisions_of(tapped[0], cuda)
        keep = Fh.deform_attention_dropout_mask(B, N, J, 8, p_drop, 17 + case, cuda) if p_drop else None
        refs = {}
        for dt in (torch.float32, torch.float64):
            r = {n: x.to(cuda, dt).requires_grad_() for n, x in t.items()}
            o = _core_reference(*(r[n] for n in NAMES), 8, 8, 0.125, keep, 1.0 / (1.0 - p_drop), masks=(m1, m2))
            (o * wo.to(cuda, dt)).sum().backward()
            refs[dt] = (o, r)
        with torch.no_grad():
            r64 = refs[torch.float64][1]
            pos = r64["gq"][None, :, None, :] - r64["vs"].view(B * 8, 1, J, 2)
            x1 = (torch.sign(pos) * torch.log(pos.abs() + 1)) @ r64["w1"].T + r64["b1"]
            x2 = torch.relu(x1) @ r64["w2"].T + r64["b2"]
            for nm, x, m in (("layer 1", x1, m1), ("layer 2", x2, m2)):
                bad = x[(x > 0) != m].abs()
                assert bad.numel() == 0 or float(bad.max()) < 2e-6, f"case {case}: a {nm} decision with |pre-activation| {float(bad.max()):.2e} differs from fp64"
        tag = f"regions case {case} ({wkind} {B}x{N}x{J} p={p_drop})"
        assert_calibrated(tag + " out", res["out"], refs[torch.float32][0], refs[torch.float64][0])
        for n in t:
            if n in ("gq", "b3"):
                continue
            assert_calibrated(tag + " d" + n, res[n], refs[torch.float32][1][n].grad, refs[torch.float64][1][n].grad)


def test_region_core_headline_shape(cuda):
    """One bag of the headline shape (100 x 100 queries, 625 keys, 8 heads) with bench.py's bias parameters and sample positions spread
    like the offsets network's: region kernels against the per-pair kernels, with the share of pairs that evaluated the MLP."""
    gen = torch.Generator().manual_seed(5)
    S, T, H = 100, 25, 8
    N, J = S * S, T * T
    t = _problem(gen, 1, N, J, H, "bench")
    ax = 2.0 * torch.arange(S, dtype=torch.float32) / (S - 1) - 1.0
    t["gq"] = torch.stack((ax.view(1, S).expand(S, S), ax.view(S, 1).expand(S, S)), dim=-1).reshape(N, 2).contiguous()
    off = torch.tanh(torch.randn(H, 2, T, T, generator=gen) * 0.7) * 4.0
    gx = torch.arange(T, dtype=torch.float32).view(1, T).expand(T, T)
    vg = torch.stack((gx, gx.t()), 0)[None] + off
    t["vs"] = (2.0 * vg / (T - 1) - 1.0).permute(0, 2, 3, 1).reshape(H, J, 2).contiguous()
    wo = torch.randn(1, N, 512, generator=gen).to(cuda)
    (a, tapped) = _run(t, cuda, wo, True, 0.1, tap=True)
    b = _run(t, cuda, wo, False, 0.1)
    rid = tapped[0]["region_ids"].to(torch.int64) & 0xFFFF
    nst = rid.shape[2] * 32
    valid = rid.view(1, H, nst // 32, J, 32).permute(0, 1, 3, 2, 4).reshape(1, H, J, nst)[..., :N]
    share = float((valid == 0xFFFF).float().mean())
    view = Fh.region_tables_view(tapped[0]["tables"])
    print(f"headline shape: {view['n_regions']} regions, {view['n_edge']} kink records, {view['n_sub']} refined cells; {share:.2e} of the pairs evaluated the MLP")
    assert view["overflow"] == 0 and share < 1e-3
    for n in a:
        if n == "b3":
            continue
        scale = max(float(b[n].abs().max()), 1e-30)
        err = float((a[n] - b[n]).abs().max()) / scale
        print(f"  {n}: {err:.2e} of its scale between the two paths")
        # d vs and the six MLP gradients are sums over up to 10 000 queries in which every pair whose pre-activation lies within fp32
        # rounding of zero may take the other branch in the other kernel family (3.2e9 decisions here: ~1e3 such ties); one flipped term
        # is ~1e-2 of a key's sum.  Both families are held to the fp64 oracle with their OWN decisions imposed (tests above, and
        # tests/test_gpu_configs.py at full size); against each other only the decision-continuous tensors are compared tightly.
        assert err <= (2e-5 if n in ("out", "q", "k", "v") else 2e-2), f"{n} differs by {err:.2e}"
