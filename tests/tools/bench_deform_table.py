"""Timing of the fused deformable-attention core alone at the headline shape (B bags x 10 000 queries x 625 keys x 8 heads, 2-D):
forward and backward of the fp32-grade path, the 16-bit MLP mode and the 16-bit table mode.  Usage: python tests/tools/bench_deform_table.py [B]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
pkg = importlib.import_module("subspace-multimodal-learning_amd")
Fh = pkg.functional
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N, J, H, G, PD = 10000, 625, 8, 8, 2
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
rn = lambda *s: torch.randn(*s, generator=g)
t = dict(q=rn(B, N, 512) * 0.4, k=rn(B, J, 512) * 0.4, v=rn(B, J, 512), vs=torch.rand(B * G, J, PD, generator=g) * 2.4 - 1.2,
         gq=torch.stack((torch.linspace(-1, 1, 100).view(1, 100).expand(100, 100), torch.linspace(-1, 1, 100).view(100, 1).expand(100, 100)), -1).reshape(N, 2).contiguous(), w1=rn(32, PD) * 0.7, b1=rn(32) * 0.3, w2=rn(32, 32) * 0.25, b2=rn(32) * 0.2,
         w3=rn(1, 32) * 0.3, b3=rn(1) * 0.1)
names = ("q", "k", "v", "vs", "gq", "w1", "b1", "w2", "b2", "w3", "b3")
d = {n: x.to(dev).requires_grad_() for n, x in t.items()}
wo = rn(B, N, 512).to(dev)
pmax = Fh.table_pmax(1.0, 1.2)
for label, kw in (("fp32-grade", {}), ("bf16 MLP", dict(compute_dtype="bf16")),
                  ("bf16 table, grid queries", dict(compute_dtype="bf16", cpb_table=True, cpb_table_pmax=pmax, cpb_table_grid=(100, 100))),
                  ("fp16 table, grid queries", dict(compute_dtype="fp16", cpb_table=True, cpb_table_pmax=pmax, cpb_table_grid=(100, 100))),
                  ("bf16 table, grid, dropout 0.1", dict(compute_dtype="bf16", cpb_table=True, cpb_table_pmax=pmax, cpb_table_grid=(100, 100), dropout_p=0.1, dropout_seed=5)),
                  ("bf16 table, any queries", dict(compute_dtype="bf16", cpb_table=True, cpb_table_pmax=pmax))):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    reps = 5
    for it in range(reps + 2):
        for x in d.values():
            x.grad = None
        ev[0].record()
        out = Fh.deform_attention(*(d[n] for n in names), heads=H, groups=G, scale=0.125, **kw)
        ev[1].record()
        out.backward(wo)
        ev[2].record()
        torch.cuda.synchronize()
        if it >= 2:
            tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
    print(f"{label:28s} forward {tf / reps:7.3f} ms   backward {tb / reps:7.3f} ms   total {(tf + tb) / reps:7.3f} ms", flush=True)
