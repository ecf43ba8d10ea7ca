"""Diagnostic: what bounds the batched 256^3 products of the Nystrom pseudo-inverse.  GPU time per launch from a captured graph
of 50 back-to-back launches (no host dispatch in the timed region), vs K, batch, epilogue, mode."""
import importlib, sys, torch
sys.path.insert(0, ".")
smml = importlib.import_module("subspace-multimodal-learning_amd")
Fh = smml.functional
dev = torch.device("cuda:0")
def t(fn, n=50, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps): g.replay()
        e1.record(s); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3
for mode, small in ((1, 1), (1, 2), (2, 1), (2, 2), (3, 1), (3, 2)):
    smml.lib().smml_gemm_set_mode(mode); smml.lib().smml_gemm_set_small_tile(small)
    for (B, M, N, K) in ((32, 256, 256, 256), (32, 256, 256, 512), (8, 256, 256, 256), (64, 256, 256, 256), (32, 512, 512, 256)):
        a = torch.randn(1, B, M, K, device=dev); b = torch.randn(1, B, K, N, device=dev); r = torch.randn(1, B, M, N, device=dev)
        out = torch.empty(1, B, M, N, device=dev)
        kw = dict(M=M, N=N, K=K, sam=K, sak=1, sbk=N, sbn=1, ldc=N, nb0=1, nb1=B, sa1=M * K, sb1=K * N, sc1=M * N)
        with torch.no_grad():
            plain = t(lambda: Fh._gemm(a, b, out, **kw))
            resid = t(lambda: Fh._gemm(a, b, out, residual=r, ldr=N, alpha=-1.0, beta=7.0, **kw))
            bt = b.transpose(-1, -2).contiguous()
            kw2 = dict(kw); kw2.update(sbk=1, sbn=K)
            nt = t(lambda: Fh._gemm(a, bt, out, **kw2))
        print(f"mode {mode} tile {(128, 64)[small - 1]}: {B} x [{M} x {K}] @ [{K} x {N}]  plain {plain:6.1f} us  +residual {resid:6.1f} us  NT {nt:6.1f} us   ({2 * B * M * N * K / plain / 1e6:6.1f} TF plain)")
smml.lib().smml_gemm_set_mode(0); smml.lib().smml_gemm_set_small_tile(0)
x = torch.randn(1 << 20, device=dev)
print(f"elementwise add on 4 MB: {t(lambda: x.add_(1.0)):.1f} us per launch in a graph")
