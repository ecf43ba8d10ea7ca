"""Diagnostic (GPU): the bilinear sampler's backward at the headline size against the oracle's formula in fp64 / fp32 (cells imposed):
error structure of d vs (systematic part)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from helpers import l2_err, rel_err, smml
from oracle.deform import bilinear_gather
Fh = smml.functional
cuda = torch.device("cuda:0")
gen = torch.Generator().manual_seed(5)
B, G, S, cg, J = 1, 8, 100, 16, 625
x = torch.randn(B, S, S, G * cg, generator=gen)
vs = torch.rand(B * G, J, 2, generator=gen) * 2.6 - 1.3
w = torch.randn(B, J, G * cg, generator=gen)
xd, vd = x.to(cuda).requires_grad_(), vs.to(cuda).requires_grad_()
kv = Fh.bilinear_sample(xd, vd, groups=G, posdim=2)
(kv * w.to(cuda)).sum().backward()
cx, cy, _ = Fh.bilinear_corners(vd.detach(), S, S, 2)
cells = (cx[:, 0].reshape(B * G, J).long(), cy[:, 0].reshape(B * G, J).long())
res = {}
for dt in (torch.float32, torch.float64):
    xr, vr = x.to(cuda, dt).requires_grad_(), vs.to(cuda, dt).requires_grad_()
    feats = xr.reshape(B, S, S, G, cg).permute(0, 3, 1, 2, 4).reshape(B * G, S, S, cg)
    ref = bilinear_gather(feats, vr[..., 0], vr[..., 1], cells).reshape(B, G, J, cg).permute(0, 2, 1, 3).reshape(B, J, G * cg)
    (ref * w.to(cuda, dt)).sum().backward()
    res[dt] = (ref.detach(), xr.grad, vr.grad)
r32, r64 = res[torch.float32], res[torch.float64]
for nm, got, i in (("kv", kv, 0), ("dx", xd.grad, 1), ("dvs", vd.grad, 2)):
    print(f"  {nm:4s} HIP max {rel_err(got, r64[i]):.2e} l2 {l2_err(got, r64[i]):.2e} | torch fp32 max {rel_err(r32[i], r64[i]):.2e} l2 {l2_err(r32[i], r64[i]):.2e}")
g64 = r64[2]
wpos = torch.rand(g64.shape, generator=torch.Generator().manual_seed(1)).to(cuda, torch.float64)
for nm, xg in (("HIP", vd.grad.double()), ("torch fp32", r32[2].double())):
    e = xg - g64
    print(f"  d vs error of {nm:10s}: along d vs {float((e * g64).sum() / (g64 * g64).sum()):+.2e}; positive-weight sum rel err {float((e * wpos).sum().abs() / (g64 * wpos).sum().abs()):.2e}; "
          f"mean err / mean |d vs| {float(e.mean() / g64.abs().mean()):+.2e}")
