"""Soak of the region kernels on random ragged shapes (N, J off the 32 / 128 tiles, J up to 2048, with and without dropout, four
parameter families): the fp32-grade region path against the per-pair MLP kernels, the 16-bit region path against the fp32-grade one, and
the global-memory region path (region_lds_cap) against the LDS-resident one.  Prints the worst relative difference per tensor class;
exits 1 on a violation of the bounds of tests/test_gpu_regions.py.   python tests/tools/soak_regions.py [cases = 40] [seed = 1]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_regions as T                 # noqa: E402  (helpers: _problem, _run)
Fh = T.Fh
cuda = torch.device("cuda", 0)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
gen = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen))
forced = [(1, 1, 768), (1, 31, 767), (2, 33, 65), (1, 257, 2), (1, 1000, 31), (3, 128, 32), (1, 129, 640), (1, 200, 900), (1, 97, 1601), (2, 64, 2048), (1, 40, 769)]
worst = {}
bad = 0
for case in range(cases + len(forced)):
    B, N, J = ri(1, 3), ri(1, 900), ri(2, 300)
    if case % 5 == 0:
        J = ri(300, 768)
    if case >= cases:
        B, N, J = forced[case - cases]
    p_drop = (0.0, 0.1, 0.25)[ri(0, 2)]
    wkind = ("random", "bench", "torch", "star")[ri(0, 3)]
    t = T._problem(gen, B, N, J, 8, wkind)
    wo = torch.randn(B, N, 512, generator=gen).to(cuda)
    a, tapped = T._run(t, cuda, wo, True, p_drop, tap=True)
    b = T._run(t, cuda, wo, False, p_drop)
    # the rigorous check (tests/test_gpu_regions.py::test_region_core_vs_fp64_with_imposed_decisions on this random shape): plain torch
    # in fp64 and fp32 with the kernels' decisions imposed on both; every kernel result within max(1e-4, 2 x the fp32 run's own error)
    m1, m2 = T.helpers.decisions_of(tapped[0], cuda)
    keep = Fh.deform_attention_dropout_mask(B, N, J, 8, p_drop, 3, cuda) if p_drop else None
    refs = {}
    for dt in (torch.float32, torch.float64):
        r = {n: x.to(cuda, dt).requires_grad_() for n, x in t.items()}
        o = T._core_reference(*(r[n] for n in T.NAMES), 8, 8, 0.125, keep, 1.0 / (1.0 - p_drop), masks=(m1, m2))
        (o * wo.to(dt)).sum().backward()
        refs[dt] = (o, r)
    for n in list(t) + ["out"]:
        if n in ("gq", "b3"):
            continue
        got = a[n]
        r32 = refs[torch.float32][0] if n == "out" else refs[torch.float32][1][n].grad
        r64 = refs[torch.float64][0] if n == "out" else refs[torch.float64][1][n].grad
        sc = max(float(r64.detach().abs().max()), 1e-30)
        if sc < 1e-9:
            continue
        e_k = float((got.double() - r64.detach()).abs().max()) / sc
        e_o = float((r32.detach().double() - r64.detach()).abs().max()) / sc
        worst[("regions vs fp64, decisions imposed", "out" if n == "out" else "grad")] = max(worst.get(("regions vs fp64, decisions imposed", "out" if n == "out" else "grad"), 0.0), e_k)
        if not e_k <= max(1e-4, 2.0 * e_o):
            # pairs WITHOUT a region (the kernels evaluate the MLP for them, in fp32; ~1e-4 of the pairs, all next to kink crossings) have no
            # exported decision: the reference takes an fp64 evaluation's.  A pre-activation of such a pair within fp32 rounding of zero can
            # be decided differently by the kernel - one pair's ReLU flip is worth ~1 / N of a key's d vs.  Reported, not counted.
            with torch.no_grad():
                ids = tapped[0]["region_ids"].view(B, 8, -1, J, 32).permute(0, 1, 3, 2, 4).reshape(B * 8, J, -1)[:, :, :N].to(torch.int64) & 0xFFFF
                none = (ids == 0xFFFF).transpose(1, 2)                              # [(B H), N, J]
                r64d = {k2: v2.detach() for k2, v2 in refs[torch.float64][1].items()}
                pos = r64d["gq"][None, :, None, :] - r64d["vs"].view(B * 8, 1, J, 2)
                x1 = (torch.sign(pos) * torch.log(pos.abs() + 1)) @ r64d["w1"].T + r64d["b1"]
                x2 = torch.relu(x1) @ r64d["w2"].T + r64d["b2"]
                tie = float(torch.minimum(x1.abs().amin(-1), x2.abs().amin(-1))[none].min()) if bool(none.any()) else float("inf")
            if tie < 3e-6:
                print(f"near-tie case {case} B{B} N{N} J{J} p{p_drop} {wkind}: {n} kernel {e_k:.3e} (an MLP-path pair has a pre-activation of {tie:.1e}: its "
                      f"decision is not exported)", flush=True)
            else:
                bad += 1
                print(f"VIOLATION case {case} vs fp64 B{B} N{N} J{J} p{p_drop} {wkind}: {n} kernel {e_k:.3e} oracle-fp32 {e_o:.3e} (smallest MLP-path pre-activation {tie:.1e})", flush=True)
    Fh.REGION_LDS_CAP = ri(8, 200)
    try:
        c = T._run(t, cuda, wo, True, p_drop)
    finally:
        Fh.REGION_LDS_CAP = 0
    mode = ("bf16", "fp16")[ri(0, 1)]
    d = T._run(t, cuda, wo, True, p_drop, compute_dtype=mode)

    def chk(tag, x, y, tols):
        global bad
        for n in x:
            sc = max(float(y[n].abs().max()), 1e-30)
            if n == "b3" or float(y[n].abs().max()) < 1e-9 or (J == 1 and n in ("q", "k", "vs", "w1", "b1", "w2", "b2", "w3")):
                continue
            err = float((x[n] - y[n]).abs().max()) / sc
            cls = "out" if n == "out" else ("qkv" if n in ("q", "k", "v") else ("vs" if n == "vs" else "mlp"))
            worst[(tag, cls)] = max(worst.get((tag, cls), 0.0), err)
            if not err <= tols[cls]:
                bad += 1
                print(f"VIOLATION case {case} {tag} B{B} N{N} J{J} p{p_drop} {wkind} {mode}: {n} {err:.3e} > {tols[cls]:.1e}", flush=True)

    # against the per-pair MLP kernels: outputs and q / k / v gradients tight; d vs and the MLP's gradients differ where the two kernel
    # families decide a ReLU at a pre-activation within rounding of zero differently (reported, bounded loosely: the fp64 check is the gate)
    chk("regions vs per-pair (fp32)", a, b, {"out": 2e-5, "qkv": 1e-4, "vs": 1e-1, "mlp": 1e-1})
    chk("global-memory regions vs LDS", c, a, {"out": 0.0, "qkv": 2e-6, "vs": 2e-5, "mlp": 2e-5})
    chk("16-bit regions vs fp32 regions", d, a, {"out": 1.5e-2, "qkv": 3e-2, "vs": 3e-2, "mlp": 6e-2})
    if case % 10 == 9:
        print(f"{case + 1} cases", flush=True)
for k in sorted(worst):
    print(f"{k[0]:34s} {k[1]:4s} worst {worst[k]:.3e}")
print("violations:", bad)
sys.exit(1 if bad else 0)
