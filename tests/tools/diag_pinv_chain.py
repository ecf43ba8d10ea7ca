"""Diagnostic: the Newton-Schulz chain (C chain, fast / generic) vs a matmul4 autograd composition vs fp64, at several m."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
smml = importlib.import_module("subspace-multimodal-learning_amd")
Fh = smml.functional
dev = torch.device("cuda:0")


def ns_ref(x, z, iters):          # any dtype, plain torch
    I = torch.eye(x.shape[-1], device=x.device, dtype=x.dtype)
    for _ in range(iters):
        xz = x @ z
        z = 0.25 * z @ (13 * I - xz @ (15 * I - xz @ (7 * I - xz)))
    return z


def ns_mm4(x, z, iters):
    for _ in range(iters):
        xz = Fh.matmul4(x, z)
        a = Fh.matmul4(xz, xz, xz, alpha=-1.0, beta=7.0)
        b = Fh.matmul4(xz, a, xz, alpha=-1.0, beta=15.0)
        z = Fh.matmul4(z, b, z, alpha=-0.25, beta=3.25)
    return z


def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())


for m in (64, 128, 256):
    gen = torch.Generator().manual_seed(m)
    a2 = torch.softmax(torch.randn(2, 3, m, m, generator=gen) * 0.5 + 4.0 * torch.eye(m), dim=-1).to(dev)
    wo = torch.randn(2, 3, m, m, generator=gen).to(dev)
    z0 = (a2.transpose(-1, -2) / (a2.abs().sum(-1).max() * a2.abs().sum(-2).max())).contiguous()
    x64 = a2.double().requires_grad_(); z64 = z0.double().requires_grad_()
    r = ns_ref(x64, z64, 6); (r * wo.double()).sum().backward()
    x32 = a2.clone().requires_grad_(); z32 = z0.clone().requires_grad_()
    r32 = ns_ref(x32, z32, 6); (r32 * wo).sum().backward()
    print(f"m={m} torch fp32 : z {rel(r32, r.detach()):.2e} dx {rel(x32.grad, x64.grad):.2e} dz0 {rel(z32.grad, z64.grad):.2e}")
    xa = a2.clone().requires_grad_(); za = z0.clone().requires_grad_()
    ra = ns_mm4(xa, za, 6); (ra * wo).sum().backward()
    print(f"m={m} matmul4 ag : z {rel(ra, r.detach()):.2e} dx {rel(xa.grad, x64.grad):.2e} dz0 {rel(za.grad, z64.grad):.2e}")
    for fast in (0, 1, 2):
        smml.lib().smml_newton_schulz_set_fast(fast)
        xc = a2.clone().requires_grad_(); zc = z0.clone().requires_grad_()
        na = importlib.import_module(smml.__name__ + ".nystrom_attention")
        rc = na._NewtonSchulz.apply(xc, zc, 6, fast == 2); (rc * wo).sum().backward()
        print(f"m={m} chain fast={fast}: z {rel(rc, r.detach()):.2e} dx {rel(xc.grad, x64.grad):.2e} dz0 {rel(zc.grad, z64.grad):.2e}")
    smml.lib().smml_newton_schulz_set_fast(-1)
