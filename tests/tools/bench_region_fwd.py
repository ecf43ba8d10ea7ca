"""Times the region forward kernel alone at the headline shape (8 bags x 8 heads x 10 000 queries x 625 keys), training mode (scores and
region ids saved) and inference mode; SMML_LIB selects a measurement variant (tests/tools/build_variants.py, -DSMML_RGN_EXP=k)."""
import os, sys, importlib, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
smml = importlib.import_module("subspace-multimodal-learning_amd")
Fh = smml.functional
dev = torch.device("cuda", 0)
B, S, T, H = 8, 100, 25, 8
N, J = S * S, T * T
gen = torch.Generator().manual_seed(5)
rn = lambda *s: torch.randn(*s, generator=gen)
shapes = {"mlp.0.0.weight": (32, 2), "mlp.0.0.bias": (32,), "mlp.1.0.weight": (32, 32), "mlp.1.0.bias": (32,), "mlp.2.weight": (1, 32), "mlp.2.bias": (1,)}
p = smml.synth.fill_params({"layer3.attn2d.rel_pos_bias." + k: v for k, v in shapes.items()}, 42, "bench")
w = [p["layer3.attn2d.rel_pos_bias." + k].to(dev) for k in shapes]
ax = 2.0 * torch.arange(S, dtype=torch.float32) / (S - 1) - 1.0
gq = torch.stack((ax.view(1, S).expand(S, S), ax.view(S, 1).expand(S, S)), dim=-1).reshape(N, 2).contiguous().to(dev)
off = torch.tanh(rn(B * H, 2, T, T) * 0.7) * 4.0
gx = torch.arange(T, dtype=torch.float32).view(1, T).expand(T, T)
vs = (2.0 * (torch.stack((gx, gx.t()), 0)[None] + off) / (T - 1) - 1.0).permute(0, 2, 3, 1).reshape(B * H, J, 2).contiguous().to(dev)
q, k, v = (rn(B, N, 512) * 0.4).to(dev), (rn(B, J, 512) * 0.4).to(dev), rn(B, J, 512).to(dev)
pmax = Fh.table_pmax(1.0, float(vs.abs().max()))
for mode in ("train", "eval"):
    qq = q.clone().requires_grad_(mode == "train")
    for it in range(6):
        if it == 2:
            Fh.TIMER.enabled = True
        out = Fh.deform_attention(qq, k, v, vs, gq, *w, heads=H, groups=H, scale=0.125, dropout_p=0.1, dropout_seed=it, cpb_regions=True, cpb_region_pmax=pmax)
    torch.cuda.synchronize()
    Fh.TIMER.enabled = False
    kt = Fh.TIMER.collect()
    print(os.environ.get("SMML_LIB", "default").split("/")[-1], mode, {k2: round(v2[1], 3) for k2, v2 in kt.items()}, flush=True)
