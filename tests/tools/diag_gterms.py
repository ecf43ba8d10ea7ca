"""Diagnostic (not a test): l2 errors of the position-bias gradients of the fused attention core against a torch fp64
evaluation, next to the errors of the same evaluation in fp32 - run once per library build to compare SMML_G_TERMS = 3 / 2.
Usage on the GPU box:  python tests/tools/diag_gterms.py [N J]"""
import importlib
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
pkg = importlib.import_module("subspace-multimodal-learning_amd")
Fh = importlib.import_module("subspace-multimodal-learning_amd.functional")
from test_gpu_parity import _core_reference  # noqa: E402

N, J = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3000, 300)
ALL_ACTIVE = len(sys.argv) > 3 and sys.argv[3] == "active"   # biases that keep every ReLU unit on: pure arithmetic error, no mask flips
cuda = torch.device("cuda:0")
gen = torch.Generator().manual_seed(7)
B, heads, groups, PD = 1, 8, 8, 2
rn = lambda *s: torch.randn(*s, generator=gen)
t = dict(q=rn(B, N, 512) * 0.4, k=rn(B, J, 512) * 0.4, v=rn(B, J, 512), vs=torch.rand(B * groups, J, PD, generator=gen) * 2.4 - 1.2,
         gq=torch.rand(N, PD, generator=gen) * 2 - 1, w1=rn(32, PD) * 0.7, b1=rn(32) * 0.3, w2=rn(32, 32) * 0.25, b2=rn(32) * 0.2,
         w3=rn(heads // groups, 32) * 0.3, b3=rn(heads // groups) * 0.1)
wo = rn(B, N, 512)
if ALL_ACTIVE:
    t['b1'] = t['b1'].abs() + 4.0
    t['b2'] = t['b2'].abs() + 30.0
if len(sys.argv) > 4 and sys.argv[4] == "small":      # keep the bias (and so the logits) O(1) although every unit is on
    t['w3'] = t['w3'] * 0.03
names = ("q", "k", "v", "vs", "gq", "w1", "b1", "w2", "b2", "w3", "b3")
dev = {n: x.to(cuda).requires_grad_() for n, x in t.items()}
out = Fh.deform_attention(*(dev[n] for n in names), heads=heads, groups=groups, scale=0.125, dropout_p=0.0, dropout_seed=0)
(out * wo.to(cuda)).sum().backward()
refs = {}
for dt in (torch.float32, torch.float64):
    r = {n: x.to(cuda, dt).requires_grad_() for n, x in t.items()}
    o = _core_reference(*(r[n] for n in names), heads, groups, 0.125)
    (o * wo.to(cuda, dt)).sum().backward()
    refs[dt] = (o, r)
l2 = lambda a, b: float((a.detach().double() - b.detach().double()).norm() / b.detach().double().norm().clamp_min(1e-300))
print(f"N={N} J={J} all-active={ALL_ACTIVE}: l2 error vs fp64          ours      torch fp32")
print(f"  out                             {l2(out, refs[torch.float64][0]):.3e}   {l2(refs[torch.float32][0], refs[torch.float64][0]):.3e}")
for n in ("w1", "b1", "w2", "b2", "w3", "vs", "q", "k", "v"):
    g64 = refs[torch.float64][1][n].grad
    print(f"  d{n:<3}                            {l2(dev[n].grad, g64):.3e}   {l2(refs[torch.float32][1][n].grad, g64):.3e}")
