"""Lists the GEMM launches of one headline training step (shape, operand layouts, split-K) with their HIP-event times."""
import importlib, os, sys, argparse
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
pkg = importlib.import_module(bench.PKG)
Fh = pkg.functional
dev = torch.device("cuda:0")
torch.manual_seed(42)
S, B, in_dim = 100, 8, 512
mil = pkg.DeformCrossTransMIL(bench.mil_args(in_dim)).to(dev).train()
path = torch.randn(B, S * S, in_dim, device=dev); omic = torch.relu(torch.randn(B, 128, device=dev)); label = torch.randint(0, 4, (B,), device=dev)
bloss = pkg.BatchLoss(B, 1)
def step():
    enc, logits, _, omic_t, vgrid = mil(path, omic)
    loss = torch.nn.functional.cross_entropy(logits, label) + torch.sum(bloss(omic_t, vgrid))
    mil.zero_grad(set_to_none=True); loss.backward()
for _ in range(2): step()
calls = []
orig = Fh._gemm
def spy(A, Bm, C, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(A, Bm, C, **kw); e1.record()
    calls.append((kw, e0, e1))
Fh._gemm = spy
step(); torch.cuda.synchronize()
tot = 0.0
for kw, e0, e1 in calls:
    ms = e0.elapsed_time(e1); tot += ms
    lay = ("A k-contig" if kw["sak"] == 1 else "A m-contig") + ", " + ("B k-contig" if kw["sbk"] == 1 and kw["sbn"] != 1 else "B n-contig")
    fl = 2.0 * kw["M"] * kw["N"] * kw["K"] * kw.get("nb0", 1) * kw.get("nb1", 1)
    print(f"M={kw['M']:6d} N={kw['N']:5d} K={kw['K']:6d} nb={kw.get('nb0',1)*kw.get('nb1',1):3d} splitk={kw.get('splitk',1):3d} {lay:26s} {ms*1e3:7.1f} us {fl/ms/1e9:6.1f} TF")
print(f"{len(calls)} GEMM launches, {tot:.3f} ms")
