#!/bin/bash
# r03 step B: headline bench of the new default build, with the split-bf16 GEMM mode, and a kernel trace
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
A="--steps 10 --warmup 3 --no-cpu-baseline --no-nystrom --no-traffic"
for mode in 0 2; do
  SMML_GEMM_MODE=$mode timeout -k 10 300 python bench.py $A > gpurun_out/bench_gm$mode.log 2>&1 || { echo "bench rc=$?"; tail -5 gpurun_out/bench_gm$mode.log; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_gm$mode.log").read().strip().splitlines()[-1])
print("gemm mode $mode: ms/step", round(d["ms_per_step"],3), "bags/s", round(d["value"],1), "cpb_bwd ms", round(d["roofline"]["avg_ms"],3), "fwd ms", round(d["roofline_fwd"]["avg_ms"],3))
PY
done
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03b -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-nystrom --no-traffic > gpurun_out/prof_r03b.log 2>&1
f=$(find gpurun_out/prof_r03b -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r03b_kernel_stats.csv; head -22 "$f" | cut -c1-150
echo done
