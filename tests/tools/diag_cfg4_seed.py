"""One full-size config-4 realisation (tests/test_gpu_configs.py::_cfg4_once) for a given seed: the parameter gradients furthest from the
fp64 oracle, beside the fp32 oracle's own distance.  Usage: [SMML_CPB_REGIONS=0] python tests/tools/diag_cfg4_seed.py [seed] [S]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from helpers import rel_err
import test_gpu_configs as T
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 23
S = int(sys.argv[2]) if len(sys.argv) > 2 else 100
cuda = torch.device("cuda", 0)
r = T._cfg4_once(cuda, 1, S, seed, host_fp32=False)
net, p32, p64 = r["net"], r["f32"][7], r["f64"][7]
rows = []
for k, p in net.named_parameters():
    if p.grad is None or getattr(p64[k], "grad", None) is None:
        continue
    rows.append((rel_err(p.grad, p64[k].grad), rel_err(p32[k].grad, p64[k].grad), k))
rows.sort(reverse=True)
for e, n, k in rows[:12]:
    print(f"{e:.3e}  fp32 oracle {n:.3e}  {k}")
k = "pathomic_net_tumor._fc1.0.weight"
d = (net.get_parameter(k).grad.double().cpu() - p64[k].grad.double().cpu()).abs()
print("worst elements of d", k, torch.topk(d.flatten(), 5), "scale", float(p64[k].grad.abs().max()))
idx = torch.topk(d.flatten(), 5).indices
print("rows / cols", [(int(i) // d.shape[1], int(i) % d.shape[1]) for i in idx])
print("row-wise max err", torch.topk(d.max(dim=1).values, 5))
