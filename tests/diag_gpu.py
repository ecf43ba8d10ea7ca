"""Diagnostic (not a test): per-parameter errors of the HIP path and of the fp32 oracle against an fp64 oracle run."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from helpers import params_for, rel_err, smml, synth
from oracle.deform import deform_cross_attention_2d

cuda = torch.device("cuda:0")
for (B, Hh, Ww) in [(1, 38, 38), (2, 12, 12), (1, 50, 50)]:
    C, N = 128, Hh * Ww
    tag = f"d2d:{B}:{Hh}:{Ww}"
    mod = smml.DeformCrossAttention2D(dim=C, dropout=0.1, grid_hw=(Hh, Ww))
    params = params_for(mod, 7, tag)
    mod.load_state_dict(params); mod = mod.to(cuda).eval()
    x1 = synth.normal((B, C, N), 7, tag + ":x1"); x2 = synth.normal((B, C, N), 7, tag + ":x2")
    w_out = synth.normal((B, C, N), 7, tag + ":wo")
    res = {}
    for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        pref = {k: v.clone().to(dt).requires_grad_() for k, v in params.items()}
        a, b = x1.clone().to(dt).requires_grad_(), x2.clone().to(dt).requires_grad_()
        o, vg = deform_cross_attention_2d(a, b, pref, grid_hw=(Hh, Ww))
        w_vg = synth.normal(tuple(vg.shape), 7, tag + ":wvg")
        ((o * w_out.to(dt)).sum() + (vg * w_vg.to(dt)).sum()).backward()
        res[name] = dict(out=o.detach(), vgrid=vg.detach(), dx1=a.grad, dx2=b.grad, **{"d" + k: v.grad for k, v in pref.items()})
    ad, bd = x1.to(cuda).requires_grad_(), x2.to(cuda).requires_grad_()
    o, vg = mod(ad, bd, return_vgrid=True)
    ((o * w_out.to(cuda)).sum() + (vg * w_vg.to(cuda)).sum()).backward()
    hip = dict(out=o.detach(), vgrid=vg.detach(), dx1=ad.grad, dx2=bd.grad, **{"d" + k: p.grad for k, p in mod.named_parameters()})
    print(f"--- B={B} grid {Hh}x{Ww}: error vs fp64 oracle (relative to tensor scale):   HIP      fp32-oracle")
    for k in hip:
        print(f"{k:42s} {rel_err(hip[k], res['f64'][k]):.2e}   {rel_err(res['f32'][k], res['f64'][k]):.2e}")
