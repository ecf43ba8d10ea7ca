"""GEMM timing (not a test): fp32-MFMA tiled kernel (mode 1) vs split-bf16 kernel (mode 2) on the shapes of the path.
Run on the GPU box: python tests/tools/bench_gemm.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from helpers import smml

Fh = smml.functional
cuda = torch.device("cuda:0")
L = smml.lib()
# (name, batch, M, N, K, a_kc, b_kc)
SHAPES = [
    # the headline step at 8 bags (80 000 tokens)
    ("8 bags fc1 / to_out fwd  x W^T", 1, 80000, 128, 512, True, True),
    ("8 bags to_q fwd          x W^T", 1, 80000, 512, 128, True, True),
    ("8 bags to_out dX         dy W", 1, 80000, 512, 128, True, False),
    ("8 bags to_q dX           dy W", 1, 80000, 128, 512, True, False),
    ("8 bags fc1 dW            dy^T x", 1, 128, 512, 80000, False, False),
    ("8 bags to_q dW           dy^T x", 1, 512, 128, 80000, False, False),
    ("8 bags fusion            x W^T", 1, 80000, 128, 128, True, True),
    ("fc1 fwd            x W^T", 1, 40000, 128, 512, True, True),
    ("fc1 dW             dy^T x", 1, 128, 512, 40000, False, False),
    ("nystrom qkv        x W^T", 1, 32768, 1536, 512, True, True),
    ("nystrom sim1       q kl^T", 64, 4096, 256, 64, True, True),
    ("nystrom attn3 v    a3 v", 64, 256, 64, 4096, True, False),
    ("nystrom a1 z       a1 z", 64, 4096, 256, 256, True, False),
    ("pinv 256^3         x z", 64, 256, 256, 256, True, False),
    ("square 4096", 1, 4096, 4096, 4096, True, False),
    # Nystrom QK / AV products at N = 10 000 (n' = 10 240), 4 bags x 8 heads
    ("nystrom N=10k q kl^T", 32, 10240, 256, 64, True, True),
    ("nystrom N=10k ql k^T", 32, 256, 10240, 64, True, True),
    ("nystrom N=10k a1 z", 32, 10240, 256, 256, True, False),
    ("nystrom N=10k a3 v", 32, 256, 64, 10240, True, False),
    ("nystrom N=10k (a1 z)(a3 v)", 32, 10240, 64, 256, True, False),
]
for name, nb, M, N, K, a_kc, b_kc in SHAPES:
    A = torch.randn(nb, M, K, device=cuda) if a_kc else torch.randn(nb, K, M, device=cuda)
    B = torch.randn(nb, N, K, device=cuda) if b_kc else torch.randn(nb, K, N, device=cuda)
    C = torch.zeros(nb, M, N, device=cuda)
    splitk = 1
    if M * N * nb < 128 * 128 * 256:
        splitk = max(1, min(K // 256, 1024 // max(1, ((M + 127) // 128) * ((N + 63) // 64) * nb)))
    kw = dict(M=M, N=N, K=K, sam=(K if a_kc else 1), sak=(1 if a_kc else M), sbk=(1 if b_kc else N), sbn=(K if b_kc else 1),
              ldc=N, nb0=nb, sa0=M * K, sb0=N * K, sc0=M * N, splitk=splitk, accumulate=1 if splitk > 1 else 0)
    ref = None
    line = f"{name:28s} b={nb:3d} {M:6d}x{N:5d}x{K:6d} splitk={splitk:3d} |"
    for mode in (1, 2):
        L.smml_gemm_set_mode(mode)
        for _ in range(2):
            C.zero_(); Fh._gemm(A, B, C, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            Fh._gemm(A, B, C, **kw)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        C.zero_(); Fh._gemm(A, B, C, **kw)
        if ref is None:
            ref = C.clone()
            err = 0.0
        else:
            err = ((C - ref).abs().max() / ref.abs().max()).item()
        line += f"  mode {mode}: {ms * 1e3:8.1f} us {2.0 * nb * M * N * K / ms / 1e9:7.1f} TF"
    L.smml_gemm_set_mode(0)
    print(line + f"   max rel diff {err:.1e}")
