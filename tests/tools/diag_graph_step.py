"""Diagnostic: can the DeformCrossTransMIL training step (forward + backward + Adam) be captured in a hipGraph as it stands?
Times eager vs replay and counts host-visible launches.  Usage on the GPU box: python tests/tools/diag_graph_step.py [bags]"""
import importlib
import sys
import time

import torch

sys.path.insert(0, ".")
pkg = importlib.import_module("subspace-multimodal-learning_amd")
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
S, in_dim = 100, 512
N = S * S
torch.manual_seed(42)
mil = pkg.DeformCrossTransMIL(bench.mil_args(in_dim))
mil.load_state_dict(pkg.synth.fill_params({k: tuple(v.shape) for k, v in mil.state_dict().items()}, 42, "bench"))
mil = mil.to(dev).train()
opt = torch.optim.Adam(mil.parameters(), lr=1e-4, weight_decay=0.1, foreach=True, capturable=True)
bloss = pkg.BatchLoss(B, 1)
path = pkg.synth.bag(B, N, in_dim, 42, "bench:bag").to(dev)
omic = torch.relu(pkg.synth.normal((B, 128), 42, "bench:omicvec")).to(dev)
label = torch.randint(0, 4, (B,), generator=torch.Generator().manual_seed(0)).to(dev)
static_loss = torch.zeros((), device=dev)


def step():
    enc, logits, _, omic_t, vgrid = mil(path, omic)
    loss = torch.nn.functional.cross_entropy(logits, label) + torch.sum(bloss(omic_t, vgrid))
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    static_loss.copy_(loss.detach())


def timeit(fn, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print(f"eager: {timeit(step):.3f} ms/step, loss {float(static_loss):.6f}")
torch.cuda.set_sync_debug_mode("error")
try:
    step()
    print("no host synchronisation inside a step")
finally:
    torch.cuda.set_sync_debug_mode("default")
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
l1 = float(static_loss)
g.replay(); torch.cuda.synchronize()
l2 = float(static_loss)
print(f"graph replay: {timeit(g.replay):.3f} ms/step, losses of two replays {l1:.6f} {l2:.6f}")
ctr = pkg.functional._REPLAY_COUNTER[0]
print("dropout replay counter after the replays:", int(ctr))
