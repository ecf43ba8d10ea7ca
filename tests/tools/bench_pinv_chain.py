"""Times the Newton-Schulz chain alone (32 problems of 256^3, 6 iterations: 24 products forward, 54 + 6 updates backward) in its three
forms.  Usage (GPU box): python tests/tools/bench_pinv_chain.py"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
smml = importlib.import_module("subspace-multimodal-learning_amd")
na = importlib.import_module(smml.__name__ + ".nystrom_attention")
dev = torch.device("cuda:0")
m = 256
a2 = torch.softmax(torch.randn(4, 8, m, m, device=dev) * 0.5 + 4.0 * torch.eye(m, device=dev), dim=-1)
z0 = (a2.transpose(-1, -2) / (a2.abs().sum(-1).max() * a2.abs().sum(-2).max())).contiguous()
wo = torch.randn(4, 8, m, m, device=dev)
for fast in (0, 1, 2):
    smml.lib().smml_newton_schulz_set_fast(fast)
    x = a2.clone().requires_grad_(); z = z0.clone().requires_grad_()
    def fwd(): return na._NewtonSchulz.apply(x, z, 6, fast == 2)
    for _ in range(3):
        out = fwd(); out.backward(wo)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    n = 20
    tf = tb = 0.0
    for _ in range(n):
        torch.cuda.synchronize(); e[0].record(); out = fwd(); e[1].record(); out.backward(wo); e[2].record(); torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    print(f"chain form {fast}: forward {tf / n * 1e3:7.1f} us ({tf / n / 24 * 1e3:5.1f} per product), backward {tb / n * 1e3:7.1f} us ({tb / n / 54 * 1e3:5.1f} per product)")
smml.lib().smml_newton_schulz_set_fast(-1)
