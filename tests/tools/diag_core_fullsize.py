"""Diagnostic (GPU): the fused attention core at the headline size (N = 10 000, J = 625) against torch fp64 / fp32 on the GPU with the
kernels' ReLU decisions imposed; error STRUCTURE of d vs (random vs systematic part: sums with positive / random-sign weights)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import helpers
from helpers import l2_err, rel_err, smml
from test_gpu_parity import _core_reference
Fh = smml.functional
cuda = torch.device("cuda:0")
gen = torch.Generator().manual_seed(3)
B, S, J, H, G, PD = 1, 100, 625, 8, 8, 2
N = S * S
rn = lambda *s: torch.randn(*s, generator=gen)
t = dict(q=rn(B, N, 512) * 0.4, k=rn(B, J, 512) * 0.4, v=rn(B, J, 512), vs=torch.rand(B * G, J, PD, generator=gen) * 2.4 - 1.2,
         gq=smml.deform_attention._build_grid_queries_2d(S, S, "cpu"), w1=rn(32, PD) * 0.7, b1=rn(32) * 0.3, w2=rn(32, 32) * 0.25, b2=rn(32) * 0.2,
         w3=rn(H // G, 32) * 0.3, b3=rn(H // G) * 0.1)
wo = rn(B, N, 512)
names = ("q", "k", "v", "vs", "gq", "w1", "b1", "w2", "b2", "w3", "b3")
dev = {n: x.to(cuda).requires_grad_(n != "gq") for n, x in t.items()}
smml.functional.DECISION_TAP = tapped = []
out = Fh.deform_attention(*(dev[n] for n in names), heads=H, groups=G, scale=0.125)
smml.functional.DECISION_TAP = None
(out * wo.to(cuda)).sum().backward()
a = tapped[0]
m1b = Fh.relu1_masks(a["vs"], a["gq"], a["w1"], a["b1"], B=B, N=N, J=J, groups=G)
m2b = Fh.relu_masks_rows(a["masks2"])[:, ::H // G].reshape(B * G, J, 2, -1)
refs = {}
CH = 1000
for dt in (torch.float32, torch.float64):
    r = {n: x.to(cuda, dt).requires_grad_(n != "gq") for n, x in t.items()}
    outs = []
    for i0 in range(0, N, CH):
        i1 = min(N, i0 + CH)
        m = (helpers.Decisions.decode(m1b, i0, i1, cuda), helpers.Decisions.decode(m2b, i0, i1, cuda))
        o = _core_reference(r["q"][:, i0:i1], r["k"], r["v"], r["vs"], r["gq"][i0:i1], r["w1"], r["b1"], r["w2"], r["b2"], r["w3"], r["b3"],
                            H, G, 0.125, masks=m)
        (o * wo[:, i0:i1].to(cuda, dt)).sum().backward()
        outs.append(o.detach())
    refs[dt] = (torch.cat(outs, 1), r)
r32, r64 = refs[torch.float32], refs[torch.float64]
print(f"out HIP {rel_err(out, r64[0]):.2e} | torch fp32 {rel_err(r32[0], r64[0]):.2e}")
for n in ("q", "k", "v", "vs", "w1", "b1", "w2", "b2", "w3"):
    g, g32, g64 = dev[n].grad, r32[1][n].grad, r64[1][n].grad
    print(f"  d{n:3s} HIP max {rel_err(g, g64):.2e} l2 {l2_err(g, g64):.2e} | torch fp32 max {rel_err(g32, g64):.2e} l2 {l2_err(g32, g64):.2e}")
# structure of the d vs error
g, g32, g64 = dev["vs"].grad.double(), r32[1]["vs"].grad.double(), r64[1]["vs"].grad
for nm, x in (("HIP", g), ("torch fp32", g32)):
    e = x - g64
    scale_bias = float((e * g64).sum() / (g64 * g64).sum())
    wpos = torch.rand(g64.shape, generator=torch.Generator().manual_seed(1)).to(cuda, torch.float64)
    wsgn = torch.randn(g64.shape, generator=torch.Generator().manual_seed(2)).to(cuda, torch.float64)
    print(f"  d vs error of {nm:10s}: component along d vs {scale_bias:+.2e}; positive-weight sum rel err {float((e * wpos).sum().abs() / (g64 * wpos).sum().abs()):.2e}; "
          f"random-sign sum {float((e * wsgn).sum().abs() / (g64 * wsgn).sum().abs()):.2e}; per-(b,g) sums {float((e.sum(1)).abs().max() / g64.sum(1).abs().max()):.2e}")
