"""Diagnostic (GPU): accuracy of the offsets network's backward in isolation (same inputs to the HIP kernels and to a torch
evaluation in fp32 / fp64): d to_offsets.2.weight, d to_offsets.0.weight / bias, dq."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from helpers import l2_err, rel_err, smml
Fh = smml.functional
cuda = torch.device("cuda:0")
torch.manual_seed(0)
B, S, G, dg = 1, 100, 8, 64
scale_w2 = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
q = torch.randn(B, S, S, G * dg) * 0.5
w0 = torch.randn(dg, 1, 6, 6) * 0.15; b0 = torch.randn(dg) * 0.1; w2 = torch.randn(2, dg) * scale_w2

def ref(dt, dvs, dvg):
    qq, a, b, c = (t.to(cuda, dt).requires_grad_() for t in (q, w0, b0, w2))
    qg = qq.reshape(B, S, S, G, dg).permute(0, 3, 4, 1, 2).reshape(B * G, dg, S, S)
    y = F.gelu(F.conv2d(qg, a, b, stride=4, padding=1, groups=dg))
    off = torch.tanh(torch.einsum("bchw,oc->bohw", y, c)) * 4.0
    th, tw = off.shape[-2:]
    gx = torch.arange(tw, dtype=dt, device=cuda).view(1, tw).expand(th, tw); gy = torch.arange(th, dtype=dt, device=cuda).view(th, 1).expand(th, tw)
    vgrid = torch.stack((gx, gy), 0) + off
    vs = torch.stack((2.0 * vgrid[:, 0] / (th - 1) - 1.0, 2.0 * vgrid[:, 1] / (tw - 1) - 1.0), dim=-1).reshape(B * G, th * tw, 2)
    ((vs * dvs.to(cuda, dt)).sum() + (vgrid * dvg.to(cuda, dt)).sum()).backward()
    return vgrid.detach(), qq.grad, a.grad, b.grad, c.grad

t = (S + 2 - 6) // 4 + 1
for kind in ("gaussian", "heavy-tailed"):
    dvs = torch.randn(B * G, t * t, 2)
    if kind == "heavy-tailed":
        dvs = dvs * torch.exp(2.5 * torch.randn(B * G, t * t, 1))
    dvg = torch.zeros(B * G, 2, t, t)
    r32, r64 = ref(torch.float32, dvs, dvg), ref(torch.float64, dvs, dvg)
    qd, a, b, c = (x.to(cuda).requires_grad_() for x in (q, w0, b0, w2))
    vg, vs = Fh.offsets(qd, a, b, c, groups=G, ks=6, r=4, posdim=2, offset_scale=4.0)
    ((vs * dvs.to(cuda)).sum() + (vg * dvg.to(cuda)).sum()).backward()
    print(f"{kind}: vgrid {rel_err(vg, r64[0]):.2e} (torch fp32 {rel_err(r32[0], r64[0]):.2e}); offsets saturation: mean 1 - tanh^2 = {float((1 - ((r64[0] - r64[0].round()) / 4).pow(2)).mean()):.3f}")
    for nm, got, i in (("dq", qd.grad, 1), ("dw0", a.grad, 2), ("db0", b.grad, 3), ("dw2", c.grad, 4)):
        print(f"  {nm:4s} HIP max {rel_err(got, r64[i]):.2e} l2 {l2_err(got, r64[i]):.2e} | torch fp32 max {rel_err(r32[i], r64[i]):.2e} l2 {l2_err(r32[i], r64[i]):.2e}")
