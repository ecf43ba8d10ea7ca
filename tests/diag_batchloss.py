"""Diagnostic: BatchLoss gradient wrt vgrid in the regime of the model (rows = identical base grid + small offsets)."""
import importlib, sys
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
smml = importlib.import_module("subspace-multimodal-learning_amd")
Fh = smml.functional
from oracle.losses import batch_loss
dev = torch.device("cuda:0")
torch.manual_seed(0)
for B, t in ((2, 25), (2, 12), (4, 25), (8, 12)):
    gx = torch.arange(t, dtype=torch.float32).view(1, t).expand(t, t); gy = torch.arange(t, dtype=torch.float32).view(t, 1).expand(t, t)
    vgrid = (torch.stack((gx, gy), 0)[None] + torch.tanh(torch.randn(B * 8, 2, t, t)) * 4.0)
    omic = torch.relu(torch.randn(B, 128) * 0.5 + 0.5)
    res = {}
    for name, dt, where in (("fp64", torch.float64, "cpu"), ("torch32cpu", torch.float32, "cpu"), ("torch32gpu", torch.float32, dev)):
        v = vgrid.detach().clone().to(where, dt).requires_grad_(); o = omic.detach().clone().to(where, dt).requires_grad_()
        l = batch_loss(o.unsqueeze(1).repeat(1, 1000, 1), v, B).sum(); l.backward()
        res[name] = (l.detach().cpu().double(), v.grad.cpu().double(), o.grad.cpu().double())
    v = vgrid.detach().clone().to(dev).requires_grad_(); o = omic.detach().clone().to(dev).requires_grad_()
    l = smml.BatchLoss(B, 1)(Fh.tile_tokens(o, 1000), v).sum(); l.backward()
    res["hip"] = (l.detach().cpu().double(), v.grad.cpu().double(), o.grad.cpu().double())
    v = vgrid.detach().clone().to(dev).requires_grad_(); o = omic.detach().clone().to(dev).requires_grad_()
    l = smml.BatchLoss(B, 1, use_tile_hint=False)(Fh.tile_tokens(o, 1000), v).sum(); l.backward()
    res["hip_nohint"] = (l.detach().cpu().double(), v.grad.cpu().double(), o.grad.cpu().double())
    ref = res["fp64"]
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))
    print(f"B={B} t={t}: loss {float(ref[0]):.4e}")
    for k in ("torch32cpu", "torch32gpu", "hip", "hip_nohint"):
        print(f"   {k:<12s} loss err {abs(float(res[k][0] - ref[0])) / abs(float(ref[0])):.2e}  dvgrid err {rel(res[k][1], ref[1]):.2e}  domic err {rel(res[k][2], ref[2]):.2e}")
