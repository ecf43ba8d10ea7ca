"""Library GEMM timing for comparison (not a test): torch.mm (rocBLAS / hipBLASLt) in fp32 on the step's shapes."""
import torch
cuda = torch.device("cuda:0")
def t(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for name, M, N, K, ta, tb in [("x W^T 80000x128x512", 80000, 128, 512, False, True), ("x W^T 80000x512x128", 80000, 512, 128, False, True),
                              ("dy W  80000x512x128", 80000, 512, 128, False, False), ("dy W  80000x128x512", 80000, 128, 512, False, False),
                              ("dy^T x 128x512x80000", 128, 512, 80000, True, False), ("dy^T x 512x128x80000", 512, 128, 80000, True, False)]:
    A = torch.randn(K, M, device=cuda).t() if ta else torch.randn(M, K, device=cuda)
    B = torch.randn(N, K, device=cuda).t() if tb else torch.randn(K, N, device=cuda)
    ms = t(lambda: torch.mm(A, B))
    print(f"{name:28s} {ms * 1e3:8.1f} us {2.0 * M * N * K / ms / 1e9:7.1f} TF")
