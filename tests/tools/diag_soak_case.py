"""One fuzz case of tests/test_gpu_deform_table.py::test_table_core_random_shapes replayed (seed, case index): d vs error of the table mode against the
fp64 interpolant, of the 16-bit MLP mode against the fp64 per-pair MLP, and of either with d bias NOT rounded to bf16 (emulated in fp64: the reference fed
with the rounded d scores) - tells rounding noise of the stored d scores from a kernel error.  Usage: python tests/tools/diag_soak_case.py 31 26"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from helpers import rel_err, smml
from test_gpu_parity import _core_reference
from test_gpu_deform_table import _interp_reference
Fh = smml.functional
seed, want = int(sys.argv[1]), int(sys.argv[2])
cuda = torch.device("cuda:0")
gen = torch.Generator().manual_seed(seed)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen))
names = ("q", "k", "v", "vs", "gq", "w1", "b1", "w2", "b2", "w3", "b3")
for case in range(want + 1):
    B, N, J = ri(1, 3), ri(1, 300), ri(2, 90)
    groups = (4, 8)[ri(0, 1)]
    heads, PD, p_drop = 8, ri(1, 2), (0.0, 0.25)[ri(0, 1)]
    rn = lambda *s: torch.randn(*s, generator=gen)
    t = dict(q=rn(B, N, 512) * 0.4, k=rn(B, J, 512) * 0.4, v=rn(B, J, 512), vs=torch.rand(B * groups, J, PD, generator=gen) * 2.4 - 1.2,
             gq=torch.rand(N, PD, generator=gen) * 2 - 1, w1=rn(32, PD) * 0.7, b1=rn(32) * 0.3, w2=rn(32, 32) * 0.25, b2=rn(32) * 0.2,
             w3=rn(heads // groups, 32) * 0.3, b3=rn(heads // groups) * 0.1)
    wo = rn(B, N, 512)
print(f"case {want}: B={B} N={N} J={J} G={groups} PD={PD} p={p_drop}")
pmax = Fh.table_pmax(1.0, 1.2)
seedd = 270 + want
keep = Fh.deform_attention_dropout_mask(B, N, J, heads, p_drop, seedd, cuda) if p_drop else None
points = smml._capi.lib().smml_deform_attn_table_points(PD)
for label, kw, ref in (("table mode vs fp64 interpolant", dict(cpb_table=True, cpb_table_pmax=pmax), "interp"), ("16-bit MLP mode vs fp64 MLP", {}, "mlp")):
    for mode in ("bf16", "fp16"):
        dev = {n: x.to(cuda).requires_grad_() for n, x in t.items()}
        out = Fh.deform_attention(*(dev[n] for n in names), heads=heads, groups=groups, scale=0.125, dropout_p=p_drop, dropout_seed=seedd, compute_dtype=mode, **kw)
        (out * wo.to(cuda)).sum().backward()
        r = {n: x.to(cuda, torch.float64).requires_grad_() for n, x in t.items()}
        if ref == "interp":
            o = _interp_reference(*(r[n] for n in names), heads, groups, 0.125, keep, 1.0 / (1.0 - p_drop), points, pmax)
        else:
            o = _core_reference(*(r[n] for n in names), heads, groups, 0.125, keep, 1.0 / (1.0 - p_drop))
        (o * wo.to(cuda, torch.float64)).sum().backward()
        g, g64 = dev["vs"].grad.double(), r["vs"].grad
        e = (g - g64).abs()
        i = int(e.argmax())
        print(f"  {label:34s} {mode}: d vs max-rel {rel_err(g, g64):.3e}  rms-rel {float(e.pow(2).mean().sqrt() / g64.abs().max()):.3e}  worst element {g.flatten()[i]:+.4e} vs {g64.flatten()[i]:+.4e}  |d vs|max {float(g64.abs().max()):.3e}")
