"""Per-step host and GPU times of the deform16 bench leg (a step that stalls on the host shows here): python tests/tools/diag_leg_steps.py [steps]"""
import gc, importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module(bench.PKG)
dev = torch.device("cuda", 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
orig = bench.timed_steps


def traced(step, n, warmup):
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    rows = []
    for i in range(steps):
        g0 = gc.get_count()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(); step(); e1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        rows.append((i, 1e3 * (t1 - t0), e0.elapsed_time(e1), g0, torch.cuda.memory_reserved() >> 20))
    for r in rows:
        print("step %2d host %7.2f ms gpu %7.2f ms gc counts %s reserved %d MiB" % r, flush=True)
    return orig(step, n, 0)


bench.timed_steps = traced
out = bench.deform16_leg(pkg, dev, 8, 100, 512, "bf16", steps=5, warmup=10)
print(out["ms_per_step"], out["timing"]["step_ms_events"])
