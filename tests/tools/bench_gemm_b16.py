"""Times the bf16-storage GEMM on the projection shapes of the Nystrom block (4 bags of 10 240 padded tokens) with HIP events.
Usage (GPU box): python tests/tools/bench_gemm_b16.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import importlib
smml = importlib.import_module("subspace-multimodal-learning_amd")
Fh = smml.functional
dev = torch.device("cuda:0")
R = 40960


def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def rnd(*s): return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)


for tile in (1, 0):
  smml.lib().smml_gemm_b16_set_tile(tile)
  print("tile mode", tile, "(1: 128 x 128 only, 0: automatic); splitk = 0: chosen by the library")
  for name, M, N, K, trans, obf, sk in [("qkv fwd  NT", R, 1536, 512, False, True, 1), ("out fwd  NT", R, 512, 512, False, False, 1),
                                        ("dx qkv   NT", R, 512, 1536, False, True, 1), ("dx out   NT", R, 512, 512, False, True, 1),
                                        ("dW qkv   TN", 1536, 512, R, True, False, 0), ("dW out   TN", 512, 512, R, True, False, 0),
                                        ("4096^3   NT", 4096, 4096, 4096, False, True, 1)]:
      if trans: a, b = rnd(K, M), rnd(K, N)
      else: a, b = rnd(M, K), rnd(N, K)
      c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16 if obf else torch.float32)
      us = timeit(lambda: Fh.gemm_b16(a, b, c, M=M, N=N, K=K, lda=a.shape[1], ldb=b.shape[1], ldc=N, trans=trans, splitk=sk))
      byts = (a.numel() + b.numel()) * 2 + c.numel() * c.element_size()
      print(f"{name}  M={M} N={N} K={K} splitk={sk}: {us:8.1f} us  {2 * M * N * K / us / 1e6:7.1f} TFLOP/s  {byts / us / 1e6:6.2f} TB/s")

print("slices of a split reduction: spread over the XCDs (0) / one XCD per slice (1)")
smml.lib().smml_gemm_b16_set_tile(-1)
for sm in (0, 1, 0, 1):
    smml.lib().smml_gemm_b16_set_slice_major(sm)
    for name, M, N, K in [("dW qkv   TN", 1536, 512, R), ("dW out   TN", 512, 512, R)]:
        a, b = rnd(K, M), rnd(K, N)
        c = torch.zeros(M, N, device=dev)
        us = timeit(lambda: Fh.gemm_b16(a, b, c, M=M, N=N, K=K, lda=M, ldb=N, ldc=N, trans=True, splitk=0))
        print(f"slice_major={sm} {name}  M={M} N={N} K={K}: {us:8.1f} us  {2 * M * N * K / us / 1e6:7.1f} TFLOP/s")
smml.lib().smml_gemm_b16_set_slice_major(1)
