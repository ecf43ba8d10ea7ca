"""Diagnostic: DeformCrossAttention2D at 100 x 100 (B = 2), gradients through `out` only / `vgrid` only, vs the fp64 oracle."""
import importlib, sys
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import helpers
from helpers import params_for, smml, synth, rel_err, l2_err
from oracle.deform import deform_cross_attention_2d
dev = torch.device("cuda:0")
B, Hh, Ww, C = 2, int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[1]) if len(sys.argv) > 1 else 100, 128
N = Hh * Ww
mod = smml.DeformCrossAttention2D(dim=C, dropout=0.1, grid_hw=(Hh, Ww))
params = params_for(mod, 17, "d2d100")
mod.load_state_dict(params); mod = mod.to(dev).eval()
ln = lambda t: torch.nn.functional.layer_norm(t.transpose(1, 2), (C,)).transpose(1, 2)
x1 = ln(synth.normal((B, C, N), 17, "d2d100:x1")); x2 = ln(torch.relu(synth.normal((B, C, N), 17, "d2d100:x2")))
wo = synth.normal((B, C, N), 17, "d2d100:wo")
for mode in ("out", "vgrid"):
    run = {}
    for dt in (torch.float32, torch.float64):
        pr = {k: v.clone().to(dt).requires_grad_() for k, v in params.items()}
        a, b = x1.clone().to(dt).requires_grad_(), x2.clone().to(dt).requires_grad_()
        o, vg = deform_cross_attention_2d(a, b, pr, grid_hw=(Hh, Ww), q_chunk=1024)
        wv = synth.normal(tuple(vg.shape), 17, "d2d100:wv")
        ((o * wo.to(dt)).sum() if mode == "out" else (vg * wv.to(dt)).sum()).backward()
        run[dt] = (a.grad, b.grad, pr)
    ad, bd = x1.to(dev).requires_grad_(), x2.to(dev).requires_grad_()
    mod.zero_grad(set_to_none=True)
    o, vg = mod(ad, bd, return_vgrid=True)
    ((o * wo.to(dev)).sum() if mode == "out" else (vg * wv.to(dev)).sum()).backward()
    r32, r64 = run[torch.float32], run[torch.float64]
    print(f"--- loss through {mode}")
    items = [("dx1", ad.grad, r32[0], r64[0])]
    if r64[1] is not None:
        items.append(("dx2", bd.grad, r32[1], r64[1]))
    for k, p in mod.named_parameters():
        if r64[2][k].grad is not None and p.grad is not None and float(r64[2][k].grad.abs().max()) > 0:
            items.append(("d" + k, p.grad, r32[2][k].grad, r64[2][k].grad))
    for name, g, g32, g64 in items:
        print(f"   {name:<40s} hip l2 {l2_err(g, g64):.2e} max {rel_err(g, g64):.2e} | oracle32 l2 {l2_err(g32, g64):.2e} max {rel_err(g32, g64):.2e}")
