"""Timing of the other section-8 rows on one MI355X (not a test): fwd + bwd milliseconds of
  a7  DeformCrossAttention1D            [B, 128, 2501] (625 keys, 2 heads per offset group)
  a9  DeformPathomicNet + BatchLoss     reference shape (B = 8, N = 2500 x 1024) and N = 10 000 x 512
  a3  TransMIL                          [B, 4096, 1024]
  a10 co-attention                      200 omic queries over 4096 path keys
Run: python tests/tools/bench_modules.py"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from helpers import smml
from test_oracle_golden import pathomic_args

dev = torch.device("cuda:0")


def timeit(name, fn, steps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    print(f"{name:64s} {(time.perf_counter() - t0) / steps * 1e3:8.2f} ms / step")


torch.manual_seed(0)
B = 8
m1 = smml.DeformCrossAttention1D(dim=128, downsample_factor=4, offset_scale=2, offset_kernel_size=6).to(dev)
x1 = torch.randn(B, 128, 2501, device=dev, requires_grad=True); x2 = torch.randn(B, 128, 2501, device=dev, requires_grad=True)
def f1():
    m1.zero_grad(set_to_none=True); x1.grad = None; x2.grad = None
    m1(x1, x2).pow(2).mean().backward()
timeit("DeformCrossAttention1D fwd+bwd, 8 x 128 x 2501", f1)

for (N, S, dim) in ((2500, 50, 1024), (10000, 100, 512)):
    net = smml.DeformPathomicNet(pathomic_args(input_path_dim=dim, batch_size=B)).to(dev).train()
    bl = smml.BatchLoss(B, 1)
    xp = torch.rand(B, N, dim, device=dev); xt = torch.randn(B, 59, device=dev); xi = torch.randn(B, 361, device=dev)
    y = torch.randint(0, 4, (B,), device=dev)
    def f2():
        net.zero_grad(set_to_none=True)
        feats, vt, vi, lg, *_ = net(x_path=xp, x_omic_tumor=xt, x_omic_immune=xi)
        loss = sum(torch.nn.functional.cross_entropy(l, y) for l in lg[:3]) + bl(lg[3], lg[4]).sum() + bl(lg[5], lg[6]).sum()
        loss.backward()
    timeit(f"DeformPathomicNet (2 branches) + 3 CE + 2 BatchLoss, {B} x {N} x {dim}", f2, steps=5, warm=2)
    if N == 10000:          # BASELINE config 4 as stated: the deformable path in bf16 compute (args.deform_compute_dtype)
        net = smml.DeformPathomicNet(pathomic_args(input_path_dim=dim, batch_size=B, deform_compute_dtype="bf16")).to(dev).train()
        timeit(f"  the same with deform_compute_dtype='bf16', {B} x {N} x {dim}", f2, steps=5, warm=2)
        for tab in ("forward", True):       # the table modes of the 16-bit core: 'forward' parity-grade, True approximate
            net = smml.DeformPathomicNet(pathomic_args(input_path_dim=dim, batch_size=B, deform_compute_dtype="bf16", deform_cpb_table=tab)).to(dev).train()
            timeit(f"  the same with deform_compute_dtype='bf16', deform_cpb_table={tab!r}, {B} x {N} x {dim}", f2, steps=5, warm=2)

args = argparse.Namespace(label_dim=4, path_dim=128, input_path_dim=1024)
tm = smml.TransMIL(args).to(dev).train()
xb = torch.randn(4, 4096, 1024, device=dev)
def f3():
    tm.zero_grad(set_to_none=True)
    enc, logits, _ = tm(xb)
    (enc.sum() + logits.pow(2).sum()).backward()
timeit("TransMIL fwd+bwd, 4 x 4096 x 1024", f3, steps=5, warm=2)
tm = smml.TransMIL(argparse.Namespace(label_dim=4, path_dim=128, input_path_dim=1024, nystrom_compute_dtype="bf16")).to(dev).train()
timeit("TransMIL fwd+bwd, 4 x 4096 x 1024, Nystrom blocks in bf16 compute mode", f3, steps=5, warm=2)

mha = smml.MultiheadAttention(embed_dim=256, num_heads=1).to(dev)
qo = torch.randn(200, 4, 256, device=dev, requires_grad=True); kp = torch.randn(4096, 4, 256, device=dev, requires_grad=True)
def f4():
    mha.zero_grad(set_to_none=True); qo.grad = None; kp.grad = None
    out, w = mha(qo, kp, kp)
    out.pow(2).mean().backward()
timeit("co-attention fwd+bwd, 200 queries x 4096 keys x 256, 4 bags", f4)
