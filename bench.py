#!/usr/bin/env python3
"""bags/sec, forward+backward(+optimizer step) of DeformCrossTransMIL on bags of 10 000 instances x 512
features (BASELINE.json metric), whole bags data-parallel over the GPUs of one node.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

One step = one training step of the reference's per-branch slice (train_test.py:57-85,188) on B synthetic bags
per GPU: DeformCrossTransMIL forward (fc1 -> fusion -> LayerNorm -> 2-D deformable cross-attention with
continuous position bias over a 100 x 100 token grid / 625 sampled keys -> pooler -> heads), cross-entropy +
BatchLoss (gathered over ranks), backward, gradient all-reduce (RCCL), Adam step.  Inputs are resident in HBM
before the timed region.  Prints ONE JSON line on rank 0.
After the timed region, at N = 1 only: `nystrom[1].kernel_counters` (per kernel of the 4 x 10 000 bf16 Nystrom leg: time, matrix-pipe busy
share, HBM bytes - three rocprofv3 --pmc child runs of `--nystrom-child`) and `roofline.traffic` from two child runs of this script under rocprofv3 --pmc (FETCH_SIZE,
WRITE_SIZE; --no-traffic skips them), the Nystrom legs (extra key `nystrom`; --no-nystrom), the CPU baseline (--no-cpu-baseline) and three more
legs of the SAME step with the fused attention core in its 16-bit compute mode (--no-deform16 skips them; none is part of `value`):
  `deform16`         bf16 compute mode, per-pair position-bias MLP (parity-grade)
  `deform16_tabfwd`  + the forward's position bias from a table of the MLP, per-pair MLP backward with mask-table decisions (parity-grade)
  `deform16_table`   forward and backward through the table (approximate: `"approximate": true`)
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "subspace-multimodal-learning_amd"

F32_MFMA_PEAK_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md: peak FP32 (matrix) = FP32 vector, dense
F16_MFMA_PEAK_TFLOPS = 2500.0       # same guide: ~2.5 PF dense BF16/F16 MFMA - the pipe every contraction of the dominant kernels issues on
CPB_BWD_MFMAS = 12                  # 32x32x16 MFMAs cpb_bwd_kernel issues per (key, 32 queries): 1 + 1 + 2 + 4 + 4
CPB_FWD_MFMAS = 7                   # deform_attn_fwd_kernel: 1 (layer 1) + 6 (layer 2, three-term split) 16-bit MFMAs per (key, 32 queries)
CPB16_BWD_MFMAS = 10                # cpb16_bwd_kernel (16-bit compute mode): 1 + 1 + 2 + 2 + 2, + 2 for d p on the matrix pipe (SMML16_DP_MFMA)
CPB16_FWD_MFMAS = 3                 # deform16_fwd_kernel: 1 (layer 1) + 2 (layer 2, one term) + 8/32 for QK^T / PV
CPB_FWD_FLOP_PER_PAIR = 2 * 2 * 32 + 2 * 32 * 32 + 2 * 32     # 2 -> 32 -> 32 -> 1 MLP = 2240
ATTN_FLOP_PER_PAIR = 2 * (2 * 64)                               # QK^T + AV per (query, key) pair and head
TRAFFIC_FILE = "r05_hbm_traffic.json"
HBM_PEAK_GBS = 8000.0               # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E ~8 TB/s peak (~6.3 TB/s achievable)


def measured_traffic(kernel, bags):
    """HBM bytes per launch of `kernel`.  bench.py cannot collect PMC counters itself: the figure is REPLAYED from the committed
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command (profiles/r02_hbm_traffic.json, gfx950 correction
    applied as MI355X_MICROARCH.md prescribes), scaled to this run's bags per launch; None if no pass is on file."""
    try:
        with open(os.path.join(ROOT, "profiles", TRAFFIC_FILE)) as f:
            t = json.load(f)
        return t["kernels"][kernel]["hbm_bytes_per_launch"] * bags / t["bags_per_launch"]
    except Exception:
        return None


def pmc_child_pass(counters, argv_tail, timeout=150):
    """ONE child run of this script under `rocprofv3 --pmc <counters> --kernel-trace` (started as a subprocess in its own process group -
    nothing is exec'd from this GPU-initialised process; killed as a group on a timeout) -> ({kernel name: {counter: [values per
    dispatch], "_ns": [durations]}}, None) or (None, reason)."""
    import csv, glob, shutil, signal, subprocess, tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    tooled = [k for k in os.environ if k.startswith(("ROCPROF", "ROCP_", "ROCTRACER", "ROCPROFILER")) or k == "HSA_TOOLS_LIB"]
    if tooled or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process runs under a profiler (" + ", ".join(tooled[:3] or ["LD_PRELOAD"]) + "): PMC child runs skipped"
    d = tempfile.mkdtemp(prefix="smml_pmc_", dir="/tmp")
    try:
        cmd = [exe, "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), *argv_tail]
        child = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
        try:
            rc = child.wait(timeout=timeout)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(child.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            child.wait()
            return None, f"CHILD_TIMEOUT: rocprofv3 --pmc {' '.join(counters)} pass exceeded {timeout} s and its process group was killed"
        if rc != 0:
            return None, (f"CHILD_SIGNAL: rocprofv3 --pmc pass died by signal {-rc}" if rc < 0 else f"rocprofv3 --pmc {' '.join(counters)} pass failed (rc {rc})")
        acc = {}
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = acc.setdefault(row["Kernel_Name"], {})
                    k.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
                    if row["Counter_Name"] == counters[0]:
                        k.setdefault("_ns", []).append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
        return acc, None
    except Exception as e:                # a profiler hiccup must not cost the bench line
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(d, ignore_errors=True)


def short_kernel_name(n):
    """Readable name of a (possibly still mangled) kernel: `attn16_fwd_kernel<bf16, float, bf16>`, `chain_bf3_dual_kernel`, ..."""
    import re
    t = {"DF16b": "bf16", "DF16_": "fp16", "f": "float"}
    m = re.match(r"_ZN12_GLOBAL__N_1\d+(\w+?_kernel)I((?:DF16b|DF16_|f)+)E", n)
    if m:
        return m.group(1) + "<" + ", ".join(t[x] for x in re.findall(r"DF16b|DF16_|f", m.group(2))) + ">"
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)


def nystrom_kernel_counters(B, n, dtype_name, budget_s=110.0):
    """Per-kernel counters of the Nystrom leg `B x n x 512 <dtype>`, collected NOW by child runs of this script
    (`--nystrom-child`) under rocprofv3: pass 1 SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE (matrix-pipe busy share of the launch's SIMD
    cycles: busy cycles / 1024 SIMDs over GUI-active cycles / 8 XCDs), passes 2 / 3 FETCH_SIZE / WRITE_SIZE (HBM bytes, separate passes,
    gfx950 correction as for `roofline.traffic`); the later passes are skipped once `budget_s` is spent.  -> dict for the bench line."""
    tail = ["--nystrom-child", f"{B},{n},{dtype_name}"]
    t0 = time.time()
    a, why = pmc_child_pass(["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"], tail)
    if a is None:
        return {"error": why}
    fe = wr = None
    notes = []
    if time.time() - t0 < budget_s * 0.45:
        fe, why_f = pmc_child_pass(["FETCH_SIZE"], tail)
        if fe is None:
            notes.append("FETCH_SIZE pass: " + why_f)
        elif time.time() - t0 < budget_s * 0.8:
            wr, why_w = pmc_child_pass(["WRITE_SIZE"], tail)
            if wr is None:
                notes.append("WRITE_SIZE pass: " + why_w)
        else:
            notes.append("WRITE_SIZE pass skipped (time budget)")
    else:
        notes.append("HBM passes skipped (time budget)")
    mean = lambda v: sum(v) / len(v)
    steps = 3
    rows = []
    for k, c in a.items():
        if not any(x in k for x in ("attn16", "gemm_b16", "gemm_f32", "chain_", "resconv", "segment_mean", "nystrom", "pinv", "softmax", "colsum")) or "_ns" not in c:
            continue
        us = mean(c["_ns"]) / 1e3
        row = {"kernel": short_kernel_name(k), "launches_per_step": len(c["_ns"]) / (2.0 * steps), "us": us,        # 3 warm-up + 3 counted executions
               "mfma_busy": mean(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / 1024.0 / max(mean(c["GRBM_GUI_ACTIVE"]) / 8.0, 1.0)}
        if fe is not None and wr is not None and k in fe and k in wr and "FETCH_SIZE" in fe[k] and "WRITE_SIZE" in wr[k]:
            hb = (2.0 * mean(fe[k]["FETCH_SIZE"]) + mean(wr[k]["WRITE_SIZE"])) * 1024.0
            row["hbm_MB"] = hb / 1e6
            row["hbm_TBps"] = hb / (us * 1e-6) / 1e12
            row["hbm_frac_of_6.3TBps"] = row["hbm_TBps"] / 6.3
        rows.append(row)
    rows.sort(key=lambda r: -r["us"] * r["launches_per_step"])
    tot = sum(r["us"] * r["launches_per_step"] for r in rows)
    for r in rows:
        r["share_of_kernel_time"] = r["us"] * r["launches_per_step"] / max(tot, 1e-9)
        if r["kernel"].startswith("attn16_") and "hbm_frac_of_6.3TBps" in r and r["mfma_busy"] > 0:
            # a QK / AV kernel should be >= 0.40 matrix-pipe busy or >= 0.70 of the achievable HBM rate
            r["meets_0.40_mfma_or_0.70_hbm"] = bool(r["mfma_busy"] >= 0.40 or r["hbm_frac_of_6.3TBps"] >= 0.70)
    chain = sum(r["us"] * r["launches_per_step"] for r in rows if r["kernel"].startswith("chain_"))
    return {"source": "live: child runs of this script under rocprofv3 --pmc (kernel trace only; the child does 3 warm-up + 3 counted steps of the leg, "
                      "both halves counted in launches_per_step), one counter group per pass" + ("; " + "; ".join(notes) if notes else ""),
            "mfma_busy_definition": "SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs, mean per dispatch",
            "kernel_time_per_step_ms": tot / 1e3, "newton_schulz_chain_ms": chain / 1e3, "newton_schulz_chain_launches": sum(
                r["launches_per_step"] for r in rows if r["kernel"].startswith("chain_")), "kernels": rows[:24], "collect_s": time.time() - t0}


def live_traffic(kernel_substr, argv_tail, timeout=150):
    """HBM bytes per launch of the kernel whose name contains `kernel_substr`, collected NOW: two child runs of this script under
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, kernel trace only, as MI355X_MICROARCH.md prescribes).
    gfx950 correction of the guide: FETCH_SIZE counts 128-byte requests as 64 bytes (x 2 for wide coalesced reads); both counters are in
    KB.  -> (bytes per launch, note) or (None, reason)."""
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        acc, why = pmc_child_pass([counter], argv_tail, timeout)
        if acc is None:
            return None, why
        v = [x for k, c in acc.items() if kernel_substr in k for x in c.get(counter, [])]
        if not v:
            return None, f"no {counter} rows for {kernel_substr}"
        vals[counter] = sum(v) / len(v)
    return vals["FETCH_SIZE"] * 1024.0 * 2.0 + vals["WRITE_SIZE"] * 1024.0, (
        f"live: two child runs of this script under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), mean over the "
        f"kernel's dispatches; FETCH_SIZE {vals['FETCH_SIZE']:.0f} KB x 2 (gfx950) + WRITE_SIZE {vals['WRITE_SIZE']:.0f} KB")


def nystrom_flop(n, D=512, h=8, d=64, m=256, iters=6, k=33):
    """forward flops F(n') of one Nystrom layer and the share of the 'QK / AV kernel' (sim1 + sim3 + attn3 v +
    (attn1 z)(attn3 v) + attn1 z) the north_star's 40 % target is quoted on; fwd + bwd = 3 x."""
    npad = -(-n // m) * m
    total = (2 * npad * D * 3 * h * d + 2 * (2 * h * npad * m * d) + 2 * h * m * m * d + iters * 5 * 2 * h * m ** 3
             + 2 * h * m * npad * d + 2 * h * npad * m * m + 2 * h * npad * m * d + 2 * k * h * npad * d + 2 * npad * D * D)
    qkav = 2 * (2 * h * npad * m * d) + 2 * h * m * npad * d + 2 * h * npad * m * m + 2 * h * npad * m * d
    return total, qkav


def mil_args(in_dim, deform_dtype=None, cpb_table=False):
    return argparse.Namespace(path_dim=128, attn_dim=2, return_vgrid=True, input_path_dim=in_dim, deform_compute_dtype=deform_dtype,
                              deform_cpb_table=cpb_table)


def algorithmic_flop_per_bag(N, J, in_dim, C=128, H=8):
    """Forward flops of one DeformCrossTransMIL branch (SURVEY.md 8(d) formula); fwd+bwd = 3x."""
    inner = 64 * H
    attn = (2 * N * (C // 8) * inner + (2 * 36 * inner * J + 2 * inner * 2 * J) + 2 * (2 * J * (C // 8) * inner)
            + 2 * (2 * H * N * J * 64) + H * N * J * CPB_FWD_FLOP_PER_PAIR + 2 * N * inner * C)
    return 2 * N * in_dim * C + 2 * N * 2 * C * C + attn


def cpu_baseline(pkg, in_dim, S, seconds_budget=90.0):
    """The oracle (plain PyTorch fp32 port of the reference's op sequence, position bias evaluated in query chunks) timed on
    this box's host cores on the workload itself: ONE bag of S*S instances (10 000 by default), forward + backward of the same
    training-step loss, all cores; one warm-up on the reference's 50 x 50 grid (pages the code in), then >= 3 timed iterations
    (fewer only if the budget runs out)."""
    from oracle.mil import deform_cross_trans_mil
    host_cores = os.cpu_count() or 1
    # PyTorch's CPU kernels for this op mix (ReLU / Linear over [pairs, 32] chunks) stop scaling near 32 threads and get SLOWER
    # beyond: measured on the 256-thread EPYC 9575F GPU box, 141.9 s per bag with all 256 threads against ~20 s with 32
    # (profiles/r02_a_bench_b8.json vs r02_b) - the baseline is given the thread count that serves it best, both counts stated
    cores = min(host_cores, int(os.environ.get("SMML_CPU_THREADS", "32")))
    torch.set_num_threads(cores)
    mil = pkg.DeformCrossTransMIL(mil_args(in_dim))
    params = pkg.synth.fill_params({k: tuple(v.shape) for k, v in mil.state_dict().items()}, 42, "bench")
    omic = torch.relu(pkg.synth.normal((1, 128), 42, "bench:omic"))

    def step(side):
        path = pkg.synth.bag(1, side * side, in_dim, 42, f"bench:cpu:{side}")
        p = {k: v.clone().requires_grad_() for k, v in params.items()}
        enc, logits, _, vg = deform_cross_trans_mil(path, omic, p, grid_hw=(side, side), q_chunk=1024)
        (torch.nn.functional.cross_entropy(logits, torch.tensor([1])) + 1e-3 * vg.pow(2).sum() + enc.sum() * 0).backward()

    step(50)                                          # warm-up
    times = []
    t_all = time.perf_counter()
    while len(times) < 3 and (time.perf_counter() - t_all) < seconds_budget:
        t0 = time.perf_counter(); step(S); times.append(time.perf_counter() - t0)
    dt = sum(times) / len(times)
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "unknown")
    except OSError:
        model = "unknown"
    return {"value": 1.0 / dt, "unit": "bags/s", "cores": cores, "host_cores": host_cores, "kind": "port", "cpu": model,
            "sample": f"oracle (plain PyTorch fp32) fwd+bwd of 1 bag of {S * S} x {in_dim} ({S}x{S} grid), {len(times)} timed iterations "
                      f"after 1 warm-up, {dt:.2f} s/bag, torch.set_num_threads({cores})"}


def timed_steps(step, steps, warmup):
    """Runs `step` warmup + steps times on the current stream.  Returns what a reader needs to tell a GPU-bound leg from a host-bound
    one: wall time per step between two synchronisations, the time the HOST needed to enqueue a step (the loop without the final
    synchronisation), and the GPU time of every single step from HIP events recorded on the launch stream (min / median / max)."""
    import gc
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    # Python's cyclic collector: a full (generation-2) pass over a process that holds a torch model walks ~10^6 objects - 50-130 ms of
    # host time, once every few hundred steps in steady state but early after the objects of model construction (it landed inside the
    # ten timed steps of one leg and doubled its wall time).  Collect now and freeze the survivors (what a training script does after
    # set-up): the timed steps then see young-generation passes only.
    gc.collect()
    gc.freeze()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(steps):
        step()
        ev[i + 1].record()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gc.unfreeze()
    per = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
    return {"ms_per_step": 1e3 * dt / steps, "steps": steps, "warmup": warmup, "gc": "collected and frozen before the timed steps",
            "step_ms_events": {"min": per[0], "median": per[len(per) // 2], "max": per[-1]},
            "host_enqueue_ms_per_step": 1e3 * t_enq / steps,
            "host_gap_ms": max(0.0, 1e3 * dt / steps - per[len(per) // 2]),
            "host_bound": bool(t_enq > 0.9 * dt)}, dt


def kernel_event_sum(kt, steps):
    """ms per step spent in the kernels the package brackets with HIP events (the fused attention family)."""
    return sum(n * ms for n, ms, _ in kt.values()) / max(steps, 1)


def nystrom_leg(pkg, dev, B, n, dtype=torch.bfloat16, steps=10, warmup=10):
    """NystromAttention(dim 512, 8 heads x 64, 256 landmarks) forward + backward on B bags of n x 512 in `dtype` (BASELINE config
    2 shape and the N = 10 000 bag; a bf16 / fp16 bag selects the block's 16-bit compute mode): ms per step from HIP events on
    the launch stream, algorithmic flops per SURVEY.md 8(d)."""
    torch.manual_seed(7)
    mod = pkg.NystromAttention(dim=512, dim_head=64, heads=8, num_landmarks=256).to(dev).eval()
    x = (torch.randn(B, n, 512, device=dev) * 0.5).to(dtype).requires_grad_()

    def step():
        mod.zero_grad(set_to_none=True); x.grad = None
        mod(x).pow(2).mean().backward()

    timing, _ = timed_steps(step, steps, warmup)
    ms = timing["step_ms_events"]["median"]
    total, qkav = nystrom_flop(n)
    tf = 3 * total * B / (ms * 1e-3) / 1e12
    pipe = mod.matrix_pipe(dtype)                                    # which matrix pipe the contractions issue on
    peak = F16_MFMA_PEAK_TFLOPS if pipe != "f32" else F32_MFMA_PEAK_TFLOPS
    return {"workload": f"NystromAttention fwd+bwd, {B} x {n} x 512 {str(dtype).replace('torch.', '')}, 256 landmarks", "ms_per_step": ms, "bags_per_s": B / (ms * 1e-3),
            "algorithmic_TFLOPs": tf, "pipe": pipe, "peak_TFLOPs": peak, "frac": tf / peak,
            "qkav_share_of_flops": qkav / total, "timing": timing}


def make_adam(params, capturable=False):
    """torch.optim.Adam as the reference configures it (lr / weight decay of config_mine.yaml:41-47 scaled down for a synthetic step); the
    fused implementation (one multi-tensor launch instead of seven) where this torch build has it - SMML_ADAM=foreach selects the other."""
    params = list(params)
    kw = dict(lr=1e-4, weight_decay=0.1, capturable=capturable)
    if os.environ.get("SMML_ADAM", "fused") == "fused":
        try:
            return torch.optim.Adam(params, fused=True, **kw)
        except (RuntimeError, TypeError, ValueError):
            pass
    return torch.optim.Adam(params, foreach=True, **kw)


def nystrom_legs(pkg, dev):
    """BASELINE config 2 (8 x 4096 bf16), the N = 10 000 bag in bf16 / fp32, and config 5 (one 50 000-instance fp16 bag)."""
    legs = []
    for B, n, dt in ((8, 4096, torch.bfloat16), (4, 10000, torch.bfloat16), (4, 10000, torch.float32), (1, 50000, torch.float16)):
        try:                                          # an extra leg must never cost the headline line
            legs.append(nystrom_leg(pkg, dev, B, n, dt))
        except Exception as e:
            legs.append({"workload": f"NystromAttention fwd+bwd, {B} x {n} x 512 {str(dt).replace('torch.', '')}", "error": f"{type(e).__name__}: {e}"})
    return legs


def deform16_leg(pkg, dev, B, S, in_dim, dtype="bf16", steps=10, warmup=10, cpb_table=False):
    """The headline training step (same model, parameters, bags, losses, Adam) with the fused attention core in its 16-bit compute
    mode (csrc/deform_attn16.hip; BASELINE config 4 names bf16): ms per step by wall clock between synchronisations, the two dominant
    kernels by HIP events on their launch stream, algorithmic flops as for the fp32-grade line.  Not part of `value`."""
    Fh = pkg.functional
    N = S * S
    torch.manual_seed(42)
    mil = pkg.DeformCrossTransMIL(mil_args(in_dim, dtype, cpb_table))
    mil.load_state_dict(pkg.synth.fill_params({k: tuple(v.shape) for k, v in mil.state_dict().items()}, 42, "bench"))
    mil = mil.to(dev).train()
    opt = make_adam(mil.parameters())
    bloss = pkg.BatchLoss(B, 1)
    path = pkg.synth.bag(B, N, in_dim, 42, "bench:bag").to(dev)
    omic = torch.relu(pkg.synth.normal((B, 128), 42, "bench:omicvec")).to(dev)
    label = torch.randint(0, 4, (B,), generator=torch.Generator().manual_seed(0)).to(dev)

    def step():
        enc, logits, _, omic_t, vgrid = mil(path, omic)
        loss = torch.nn.functional.cross_entropy(logits, label) + torch.sum(bloss(omic_t, vgrid))
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    last = {}

    def one():
        last["loss"] = step()

    for _ in range(warmup):
        one()
    torch.cuda.synchronize()
    Fh.TIMER.enabled = True
    timing, dt = timed_steps(one, steps, 0)
    Fh.TIMER.enabled = False
    kt = Fh.TIMER.collect()
    loss = last["loss"]
    timing["warmup"] = warmup
    timing["package_kernel_events_ms_per_step"] = kernel_event_sum(kt, steps)
    out = {"workload": f"the headline step with DeformCrossAttention2D(compute_dtype='{dtype}'): {B} bags of {N} x {in_dim} per step, "
                       f"fwd + bwd + Adam, CE + BatchLoss; parameters, inputs, outputs and gradients fp32 in memory",
           "dtype": dtype, "steps": steps, "ms_per_step": 1e3 * dt / steps, "bags_per_s": B * steps / dt, "loss_finite": bool(torch.isfinite(loss).item()),
           "timing": timing}
    if cpb_table == "forward":
        out["workload"] += ("; the FORWARD takes the position bias from a table of the MLP (cpb_table='forward': evaluated once per call on a 96 x 96 grid, "
                            "interpolated per pair, |error| <= ~1e-3 of the bias range); the backward differentiates the per-pair MLP itself (layer-2 decisions from a 1024 x 1024 mask table): "
                            "parity-grade at the 16-bit mode's tolerances (tests/test_gpu_deform16.py, tabfwd cases)")
        if "deform_table_fwd" in kt:
            n, ms, pairs = kt["deform_table_fwd"]
            out["deform_table_fwd"] = {"avg_ms": ms, "launches": n, "pairs_per_launch": pairs}
    elif cpb_table:
        out["workload"] += ("; position bias in TABLE mode (cpb_table=True: the MLP evaluated once per call on a 96 x 96 grid, interpolated per pair - an "
                            "APPROXIMATION of the reference's per-pair MLP, see tests/test_gpu_deform_table.py; not a parity-grade line)")
        out["approximate"] = True
        for key in ("deform_table_fwd", "cpb_table_bwd"):
            if key in kt:
                n, ms, pairs = kt[key]
                out[key] = {"avg_ms": ms, "launches": n, "pairs_per_launch": pairs,
                            "note": "forward kernel" if key == "deform_table_fwd" else "d vs (per pair) + d table (two dense products per key on the matrix pipe)"}
    out["kernel_events"] = {k: {"launches": v[0], "avg_ms": v[1], "pairs_per_launch": v[2]} for k, v in kt.items()}
    if "deform16_region_fwd" in kt:
        # the region form of the 16-bit core (the default wherever it applies): the same fp32 lookup of the position bias as the headline
        # path, single-term 16-bit operands on the matrix pipe, fp16 scores (2 B) + region ids (2 B) saved per pair, bf16 d scores
        out["workload"] += "; position bias per linear region of its MLP (the headline path's fp32 lookup, csrc/cpb_regions.h)"
        n, ms, pairs = kt["deform16_region_fwd"]
        Hh = 8
        abytes = pairs * 4 + B * Hh * N * (512 + 4) + B * Hh * (pairs // (B * Hh * N)) * (512 + 8)
        out["roofline_fwd"] = {"kernel": "deform_region_fwd_kernel<true, bf16>", "bound": "hbm", "achieved": abytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": abytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_ms": ms, "launches": n,
                               "algorithmic_bytes_per_launch": abytes, "limiter": "vector issue"}
        if "cpb16_region_bwd" in kt:
            n2, ms2, pairs2 = kt["cpb16_region_bwd"]
            b2 = pairs2 * 4
            out["roofline"] = {"kernel": "cpb_region_bwd_kernel<bf16 d scores>", "bound": "hbm", "achieved": b2 / (ms2 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": b2 / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_ms": ms2, "launches": n2,
                               "algorithmic_bytes_per_launch": b2, "limiter": "vector issue + LDS atomics"}
    if "cpb16_bwd" in kt:
        n, ms, pairs = kt["cpb16_bwd"]
        flop = pairs * 2 * CPB_FWD_FLOP_PER_PAIR
        out["roofline"] = {"kernel": "cpb16_bwd_kernel<2>", "bound": "mfma", "achieved": flop / (ms * 1e-3) / 1e12, "peak": F16_MFMA_PEAK_TFLOPS,
                           "unit": "TFLOP/s", "frac": flop / (ms * 1e-3) / 1e12 / F16_MFMA_PEAK_TFLOPS, "avg_ms": ms, "launches": n,
                           "flop_per_launch": flop, "mfmas_per_key_and_32_queries": CPB16_BWD_MFMAS}
    if "deform16_fwd" in kt:
        n, ms, pairs = kt["deform16_fwd"]
        flop = pairs * (CPB_FWD_FLOP_PER_PAIR + ATTN_FLOP_PER_PAIR)
        out["roofline_fwd"] = {"kernel": "deform16_fwd_kernel<2, true>", "bound": "mfma", "achieved": flop / (ms * 1e-3) / 1e12,
                               "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flop / (ms * 1e-3) / 1e12 / F16_MFMA_PEAK_TFLOPS,
                               "avg_ms": ms, "launches": n, "flop_per_launch": flop, "mfmas_per_key_and_32_queries": CPB16_FWD_MFMAS}
    del mil, opt, bloss, path, omic, label
    torch.cuda.empty_cache()
    return out


def dp_overhead_child(pkg, dev, B, S, in_dim, steps, warmup):
    """Child-process body of the `data_parallel_at_world_1` leg: the headline step plain and wrapped in BagDataParallel over a ONE-rank
    RCCL group (every bucket all-reduced with ReduceOp.AVG from the gradient hooks while backward runs; the average over one rank is
    the identity), A/B in one process on one box.  Prints one JSON object."""
    import datetime
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=90))
    N = S * S
    res = {}
    for name in ("plain", "wrapped", "wrapped_2MiB_buckets", "plain_again"):
        torch.manual_seed(42)
        mil = pkg.DeformCrossTransMIL(mil_args(in_dim))
        mil.load_state_dict(pkg.synth.fill_params({k: tuple(v.shape) for k, v in mil.state_dict().items()}, 42, "bench"))
        mil = mil.to(dev).train()
        model = mil
        if name.startswith("wrapped"):
            model = pkg.BagDataParallel(mil, collectives_at_world_1=True, **({"bucket_bytes": 2 << 20} if "2MiB" in name else {}))
        opt = make_adam(mil.parameters())
        bloss = pkg.BatchLoss(B, 1)
        path = pkg.synth.bag(B, N, in_dim, 42, "bench:bag").to(dev)
        omic = torch.relu(pkg.synth.normal((B, 128), 42, "bench:omicvec")).to(dev)
        label = torch.randint(0, 4, (B,), generator=torch.Generator().manual_seed(0)).to(dev)

        def step():
            enc, logits, _, omic_t, vgrid = model(path, omic)
            loss = torch.nn.functional.cross_entropy(logits, label) + torch.sum(bloss(omic_t, vgrid))
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()

        timing, _ = timed_steps(step, steps, warmup)
        res[name] = timing
        if name.startswith("wrapped"):
            res["reducer" + name[len("wrapped"):]] = {k: model.stats.get(k) for k in ("buckets", "launched_in_backward", "skipped", "hook_host_ms")} | model.timing()
        del step, model, mil, opt, bloss, path, omic, label
        torch.cuda.empty_cache()
    dist.destroy_process_group()
    # medians of the per-step event times: one slow step (a watchdog tick, the first collective) would otherwise decide a 10-step mean
    med = lambda k: res[k]["step_ms_events"]["median"]
    res["overhead_vs_plain"] = med("wrapped") / (0.5 * (med("plain") + med("plain_again"))) - 1.0
    res["overhead_vs_plain_2MiB_buckets"] = med("wrapped_2MiB_buckets") / (0.5 * (med("plain") + med("plain_again"))) - 1.0
    res["overhead_vs_plain_wall"] = res["wrapped"]["ms_per_step"] / (0.5 * (res["plain"]["ms_per_step"] + res["plain_again"]["ms_per_step"])) - 1.0
    print(json.dumps(res))


def dp_overhead_leg(pkg, dev, B, S, in_dim, headline_ms, timeout=240):
    """Runs dp_overhead_child in a child process (subprocess, nothing exec'd from this GPU-initialised process): an RCCL
    initialisation that hangs can then cost this leg, not the line."""
    import signal, subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--dp-child", "--bags", str(B), "--grid", str(S), "--in-dim", str(in_dim), "--steps", "10", "--warmup", "10"]
    child = subprocess.Popen(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, start_new_session=True, text=True)
    try:
        so, _ = child.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(child.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        child.wait()
        return {"error": f"CHILD_TIMEOUT: the one-rank RCCL child exceeded {timeout} s and its process group was killed"}
    if child.returncode != 0:
        return {"error": f"child exited with code {child.returncode}"}
    res = json.loads(so.strip().splitlines()[-1])
    res["what"] = ("the headline step wrapped in BagDataParallel over a one-rank RCCL group (collectives_at_world_1): every bucket all-reduced with ReduceOp.AVG from "
                   "the gradient hooks while backward runs; `overhead_vs_plain` = median step of wrapped / mean(median step of plain before, after) - 1 (HIP events per step), all three in one child process; `_wall` = the same from the 10-step wall means")
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bags", type=int, default=8, help="bags per GPU per step (the reference trains with batch_size 8 per rank, config/config_mine.yaml:38)")
    ap.add_argument("--grid", type=int, default=100, help="token grid side (N = grid^2)")
    ap.add_argument("--in-dim", type=int, default=512, help="bag feature width")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--deform-table", nargs="?", const=True, default=False, help="measurement switch (with --deform-dtype): the headline step with the tabulated position "
                    "bias (approximate mode; `--deform-table forward`: table in the forward only, exact backward; the default run reports both in the extra keys `deform16_table` / `deform16_tabfwd`)")
    ap.add_argument("--no-deform16", action="store_true", help="skip the 16-bit-compute-mode leg of the headline step (extra key `deform16`, not part of `value`)")
    ap.add_argument("--no-nystrom", action="store_true", help="skip the Nystrom legs (extra key `nystrom`, not part of `value`)")
    ap.add_argument("--no-dp-overhead", action="store_true", help="skip the leg that wraps the headline step in a one-rank RCCL group (extra key `data_parallel_at_world_1`)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc child runs that measure `roofline.traffic`")
    ap.add_argument("--no-nystrom-pmc", action="store_true", help="skip the rocprofv3 --pmc child runs behind `nystrom[1].kernel_counters`")
    ap.add_argument("--nystrom-child", default=None, help="internal: B,n,dtype - run that Nystrom leg alone (3 + 3 steps) and exit (profiled by the parent)")
    ap.add_argument("--deform-dtype", default=None, choices=[None, "bf16", "fp16"],
                    help="measurement switch: run the HEADLINE step itself with the fused attention core in its 16-bit compute mode (the "
                         "default line stays fp32-grade; the default run reports the 16-bit step in the extra key `deform16`)")
    ap.add_argument("--graph", action="store_true", help="one GPU only: capture the step in a hipGraph and time replays (the per-kernel "
                    "HIP-event times of `roofline` then come from the eager warm-up steps)")
    ap.add_argument("--dp-child", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.dp_child:
        torch.cuda.set_device(0)
        dp_overhead_child(importlib.import_module(PKG), torch.device("cuda", 0), a.bags, a.grid, a.in_dim, a.steps, a.warmup)
        return
    if a.nystrom_child:
        torch.cuda.set_device(0)
        nb, nn_, ndt = a.nystrom_child.split(",")
        nystrom_leg(importlib.import_module(PKG), torch.device("cuda", 0), int(nb), int(nn_), getattr(torch, ndt), steps=3, warmup=3)
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # rehearsal hooks for a one-GPU box (the N > 1 path with every rank on cuda:0 over gloo; RCCL refuses two ranks per
    # device): SMML_BENCH_ONE_DEVICE=1 SMML_DIST_BACKEND=gloo.  Never set by the driver.
    if os.environ.get("SMML_BENCH_ONE_DEVICE") == "1":
        local = 0
    backend = os.environ.get("SMML_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"

    pkg = importlib.import_module(PKG)
    Fh = pkg.functional
    S, B, in_dim = a.grid, a.bags, a.in_dim
    N = S * S
    J = pkg.lib().smml_offsets_out_len(S, 6, 4) ** 2

    torch.manual_seed(42)
    if a.deform_table and not a.deform_dtype:
        raise SystemExit("--deform-table needs --deform-dtype bf16|fp16")
    mil = pkg.DeformCrossTransMIL(mil_args(in_dim, a.deform_dtype, a.deform_table))
    mil.load_state_dict(pkg.synth.fill_params({k: tuple(v.shape) for k, v in mil.state_dict().items()}, 42, "bench"))
    mil = mil.to(dev).train()              # train mode: attention dropout 0.1 (DeformCrossTransMIL.py:49) is active
    model = pkg.BagDataParallel(mil) if world > 1 else mil
    use_graph = bool(a.graph and world == 1)
    opt = make_adam(mil.parameters(), capturable=use_graph)
    bloss = pkg.BatchLoss(B, world)
    # synthetic bags, resident in HBM before timing; a different bag set per rank (whole bags per rank)
    path = pkg.synth.bag(B, N, in_dim, 42 + rank, "bench:bag").to(dev)
    omic = torch.relu(pkg.synth.normal((B, 128), 42 + rank, "bench:omicvec")).to(dev)
    label = torch.randint(0, 4, (B,), generator=torch.Generator().manual_seed(rank)).to(dev)

    def step():
        enc, logits, _, omic_t, vgrid = model(path, omic)
        loss = torch.nn.functional.cross_entropy(logits, label) + torch.sum(bloss(omic_t, vgrid))
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if use_graph:
        # eager steps first (they also create the dropout replay counter and give the per-kernel event times), then one capture;
        # every replay draws new dropout masks through the device-resident seed offset (include/smml.h)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            Fh.TIMER.enabled = True
            for _ in range(max(a.warmup, 3)):
                step()
            Fh.TIMER.enabled = False
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                step()
        torch.cuda.current_stream().wait_stream(side)
        graph.replay()
        fence()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            graph.replay()
        fence()
        dt = time.perf_counter() - t0
    else:
        for _ in range(a.warmup):
            step()
        import gc
        gc.collect()                       # (see timed_steps: no full pass of Python's cyclic collector over the model's objects inside the timed steps)
        gc.freeze()
        fence()
        Fh.TIMER.enabled = True
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        Fh.TIMER.enabled = False
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    kt = Fh.TIMER.collect()
    if a.deform_dtype:                      # measurement switch: the 16-bit kernels report under the same two roofline keys
        kt = {{"deform16_fwd": "deform_attn_fwd", "cpb16_bwd": "cpb_bwd", "deform_table_fwd": "deform_attn_fwd", "cpb_table_bwd": "cpb_bwd"}.get(k, k): v
              for k, v in kt.items()}
    dp_info = None
    if world > 1:
        # self-check of a multi-GPU run: the world size the collective backend reports, every rank's device, and the data-parallel
        # wrapper's counters of the LAST step (buckets launched while backward was still running, buckets skipped as grad-less,
        # host time spent inside the gradient hooks)
        st = dict(getattr(model, "stats", {}))
        st.update(model.timing() if hasattr(model, "timing") else {})
        mine = {"rank": rank, "exposed_wait_ms": st.get("exposed_wait_ms"), "allreduce_span_ms": st.get("allreduce_span_ms"),
                "hidden_share": st.get("hidden_share"), "device": torch.cuda.get_device_name(dev), "device_index": local, "buckets": st.get("buckets"),
                "launched_in_backward": st.get("launched_in_backward"), "skipped": st.get("skipped"),
                "hook_host_ms_per_step": st.get("hook_host_ms")}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        dp_info = {"backend": dist.get_backend(), "world_size_seen": dist.get_world_size(), "ranks": gathered}
    if rank == 0:
        out = {
            "metric": "bags/sec fwd+bwd, DeformCrossTransMIL N=10k x 512",
            "value": world * B * a.steps / dt, "unit": "bags/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": (a.deform_dtype + (" (fused attention core" + (", forward bias from a table" if a.deform_table == "forward" else (", tabulated position bias: approximate" if a.deform_table else "")) + "); f32 elsewhere")) if a.deform_dtype else "f32",
            "data": "synthetic",
            "config": {"workload": f"DeformCrossTransMIL training step (fwd + bwd + Adam), bag {N} x {in_dim} fp32, "
                                   f"token grid {S}x{S}, {J} sampled keys, path_dim 128, heads 8, CE + BatchLoss",
                       "bags_per_gpu": B, "global_batch": B * world, "instances": N, "feature_dim": in_dim,
                       "parallelism": f"dp{world} (whole bags per rank, RCCL gradient all-reduce)" + (", step replayed from one hipGraph" if use_graph else "")},
        }
        out["kernel_events"] = {k: {"launches": v[0], "avg_ms": v[1], "pairs_per_launch": v[2]} for k, v in kt.items()}   # HIP events on the launch stream
        H = 8
        regions = "deform_region_fwd" in kt            # the position bias per linear region of its MLP (csrc/cpb_regions.h): the default fp32-grade path
        if regions:
            # Dominant kernel: the fused forward.  With the per-pair MLP gone it is a memory kernel: per (query, key) pair and head it writes
            # the pre-softmax score (4 B, saved for the backward, the dropout decision in its lowest bit) and the pair's region id (2 B);
            # per (bag, head): q read + out written (2 x 256 B per query, + 4 B lse), k + v + sample positions read (520 B per key).
            n, ms, pairs = kt["deform_region_fwd"]
            abytes = pairs * 6 + B * H * N * (512 + 4) + B * H * J * (512 + 8)
            ach = abytes / (ms * 1e-3) / 1e9
            flop = pairs * (CPB_FWD_FLOP_PER_PAIR + ATTN_FLOP_PER_PAIR)
            out["roofline"] = {"kernel": "deform_region_fwd_kernel<true>", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": ach / HBM_PEAK_GBS, "traffic": None, "traffic_source": "not collected (--no-traffic, or N > 1)",
                               "launches": n, "avg_ms": ms, "algorithmic_bytes_per_launch": abytes,
                               "reference_flops_view": {"flop_per_launch": flop, "achieved_TFLOPs": flop / (ms * 1e-3) / 1e12,
                                                        "frac_of_16bit_mfma_peak": flop / (ms * 1e-3) / 1e12 / F16_MFMA_PEAK_TFLOPS,
                                                        "note": "the reference's op count (2496 flop per pair: bias MLP 2240 + QK^T / PV 256) over this kernel's time - "
                                                                "the kernel no longer executes the MLP's flops (one table lookup + 2 FMAs per pair)"},
                               "note": "achieved = algorithmic bytes (6 B per pair: saved score + region id; q / k / v / out once) / HIP-event time of the kernel, "
                                       "against the HBM peak of MI355X_MICROARCH.md (8 TB/s; ~6.3 TB/s achievable).  What bounds it today is the issue of ~90 vector "
                                       "instructions per pair (signed logs, cell index, kink record, region lookup, dropout hash, softmax, fp16 splits) at two "
                                       "waves per SIMD (256 registers each), not HBM and no longer the latency of the table gathers "
                                       "(profiles/r05_pmc_summary.txt, DESIGN.md section 5).",
                               "limiter": "vector issue"}
            if "cpb_region_bwd" in kt:
                n2, ms2, pairs2 = kt["cpb_region_bwd"]
                b2 = pairs2 * 6 + B * H * J * 8 * 2
                out["roofline_bwd"] = {"kernel": "cpb_region_bwd_kernel", "bound": "hbm", "achieved": b2 / (ms2 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": b2 / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, "launches": n2, "avg_ms": ms2, "algorithmic_bytes_per_launch": b2,
                                       "note": "position-bias backward per linear region: reads d score (4 B) + region id (2 B) per pair, three 64-bit integer LDS adds per "
                                               "run of pairs in one region; bound by vector issue (fp64 conversions of the run flushes) and LDS atomics, not by HBM",
                                       "limiter": "vector issue + LDS atomics"}
        elif "cpb_bwd" in kt:
            n, ms, pairs = kt["cpb_bwd"]
            flop = pairs * 2 * CPB_FWD_FLOP_PER_PAIR           # backward = 2x forward flops (4480 per pair), recompute not counted
            ach = flop / (ms * 1e-3) / 1e12
            issued = (pairs / 32.0) * CPB_BWD_MFMAS * 32768 / (ms * 1e-3) / 1e12   # what the kernel executes: 12 MFMAs of 2 * 16384 flop
            out["roofline"] = {"kernel": "cpb_bwd_kernel<2>", "bound": "mfma", "achieved": ach, "peak": F16_MFMA_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": ach / F16_MFMA_PEAK_TFLOPS, "traffic": None, "traffic_source": "not collected",
                               "launches": n, "avg_ms": ms, "flop_per_launch": flop,
                               "issued_16bit": {"achieved": issued, "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": issued / F16_MFMA_PEAK_TFLOPS},
                               "note": "per-pair MLP kernels (SMML_CPB_REGIONS=0 or a 16-bit mode): algorithmic flops (4480 per pair and head, SURVEY 8(d)) / HIP-event "
                                       "time, against the dense 16-bit MFMA peak; the kernel is vector-issue bound (~265 VALU per 12 MFMAs)"}
            if "deform_attn_fwd" in kt:
                n, ms, pairs = kt["deform_attn_fwd"]
                flop = pairs * (CPB_FWD_FLOP_PER_PAIR + ATTN_FLOP_PER_PAIR)
                out["roofline_fwd"] = {"kernel": "deform_attn_fwd_kernel<2, true>", "bound": "mfma", "achieved": flop / (ms * 1e-3) / 1e12,
                                       "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flop / (ms * 1e-3) / 1e12 / F16_MFMA_PEAK_TFLOPS,
                                       "launches": n, "avg_ms": ms, "flop_per_launch": flop}
        if world > 1:
            out["data_parallel"] = dp_info               # what the collective backend saw + per-rank overlap counters: a SCALE run checks itself
        # the headline is measured: persist it before anything else can go wrong (an extra leg must never cost the line)
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "bench_headline.json"), "w") as f:
                f.write(json.dumps(out) + "\n")
        except OSError:
            pass
        print("HEADLINE " + json.dumps({k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "ms_per_step")}), file=sys.stderr, flush=True)
        # everything below runs on a GPU the headline model no longer occupies
        del step, model, mil, opt, bloss, path, omic, label
        if use_graph:
            del graph
        torch.cuda.empty_cache()
        if world == 1 and not a.no_dp_overhead and not a.deform_dtype:
            try:
                out["data_parallel_at_world_1"] = dp_overhead_leg(pkg, dev, B, S, in_dim, out["ms_per_step"])
            except Exception as e:
                out["data_parallel_at_world_1"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not a.no_deform16 and not a.deform_dtype:
            # BASELINE config 4 as stated (bf16 compute) and its table variants: the same step with the fused core in its 16-bit mode
            for key, kw in (("deform16", {}), ("deform16_tabfwd", {"cpb_table": "forward"}), ("deform16_table", {"cpb_table": True})):
                try:
                    out[key] = deform16_leg(pkg, dev, B, S, in_dim, "bf16", steps=max(5, min(a.steps, 20)), **kw)
                    out[key]["speedup_vs_fp32_line"] = out[key]["bags_per_s"] / out["value"]
                except Exception as e:
                    out[key] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not a.no_nystrom:
            # the north_star's Nystrom target, driver-run: BASELINE config 2 shape and the N = 10 000 bag (not part of `value`);
            # timed BEFORE the PMC child runs below so that nothing of theirs can still be on the GPU
            out["nystrom"] = nystrom_legs(pkg, dev)
            if not a.no_nystrom_pmc and len(out["nystrom"]) > 1 and "error" not in out["nystrom"][1]:
                try:                                  # per-kernel matrix-pipe busy share and HBM bytes of the 4 x 10 000 bf16 leg, live
                    out["nystrom"][1]["kernel_counters"] = nystrom_kernel_counters(4, 10000, "bfloat16")
                except Exception as e:
                    out["nystrom"][1]["kernel_counters"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and "roofline" in out:
            kname = "deform_region_fwd_kernel" if regions else "cpb_bwd_kernel"
            traffic, tsrc = None, "not collected (--no-traffic)"
            if not a.no_traffic:
                traffic, tsrc = live_traffic(kname, ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-nystrom", "--no-traffic", "--no-deform16",
                                                     "--no-dp-overhead", "--bags", str(B), "--grid", str(S), "--in-dim", str(in_dim)])
                if tsrc.startswith(("CHILD_TIMEOUT", "CHILD_SIGNAL")):
                    out["pmc_child_failed"] = tsrc           # a hung / crashed profiler child is surfaced, not folded into a fallback
                if traffic is None and (S, in_dim) == (100, 512):
                    traffic = measured_traffic(kname, B)
                    tsrc = f"replayed from profiles/{TRAFFIC_FILE} (rocprofv3 --pmc passes of this command, committed); live collection: {tsrc}"
            out["roofline"]["traffic"], out["roofline"]["traffic_source"] = traffic, tsrc
        if world == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(pkg, in_dim, S)
            except Exception as e:                    # the GPU figures above are measured: a host-side failure must not lose the line
                out["cpu_baseline"] = {"value": None, "unit": "bags/s", "cores": 0, "kind": "port", "sample": f"failed: {type(e).__name__}: {e}"}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
