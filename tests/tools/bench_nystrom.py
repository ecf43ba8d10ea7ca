"""Timing of the Nystrom block (BASELINE configs 2 / 4-size): NystromAttention(dim 512, 8 heads x 64, 256 landmarks)
forward + backward on [B, n, 512] fp32.  Prints one JSON line; the contraction flops follow SURVEY.md 8(d)."""
import argparse, importlib, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
smml = importlib.import_module("subspace-multimodal-learning_amd")


def nystrom_fwd_flop(n, D=512, h=8, d=64, m=256, iters=6, k=33):
    npad = -(-n // m) * m
    return (2 * npad * D * 3 * h * d + 2 * (2 * h * npad * m * d) + 2 * h * m * m * d + iters * 5 * 2 * h * m ** 3
            + 2 * h * m * npad * d + 2 * h * npad * m * m + 2 * h * npad * m * d + 2 * k * h * npad * d + 2 * npad * D * D)


ap = argparse.ArgumentParser(); ap.add_argument("--n", type=int, default=4096); ap.add_argument("--bags", type=int, default=4)
ap.add_argument("--steps", type=int, default=10); ap.add_argument("--dtype", default="float32")
ap.add_argument("--graph", action="store_true", help="capture the step in a hipGraph and time replays"); a = ap.parse_args()
dev = torch.device("cuda:0")
mod = smml.NystromAttention(dim=512, dim_head=64, heads=8, num_landmarks=256).to(dev).eval()
x = (torch.randn(a.bags, a.n, 512, device=dev) * 0.5).to(getattr(torch, a.dtype)).requires_grad_()
def step():
    mod.zero_grad(set_to_none=True); x.grad = None
    mod(x).pow(2).mean().backward()
run = step
if a.graph:
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): step()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            step()
    torch.cuda.current_stream().wait_stream(side)
    run = g.replay
for _ in range(3): run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps): run()
t_issue = (time.perf_counter() - t0) / a.steps        # host time to enqueue a step (equals the step time when the step is host-bound)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
fl = 3 * nystrom_fwd_flop(a.n) * a.bags
print(json.dumps({"workload": f"NystromAttention fwd+bwd, {a.bags} x {a.n} x 512 {a.dtype}, 256 landmarks", "ms_per_step": dt * 1e3,
                  "bags_per_s": a.bags / dt, "algorithmic_TFLOPs": fl / dt / 1e12, "pipe": mod.matrix_pipe(x.dtype), "graph": bool(a.graph), "host_issue_ms": t_issue * 1e3}))
