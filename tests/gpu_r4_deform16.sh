#!/bin/bash
# first GPU trip of the 16-bit deformable-attention mode: parity tests, then step time fp32-grade vs bf16 vs fp16 on one box
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_deform16.py -x -q -s -m gpu > gpurun_out/r4_deform16_tests.log 2>&1
rc=$?
tail -40 gpurun_out/r4_deform16_tests.log
[ $rc -ne 0 ] && echo "tests rc=$rc"
for m in none bf16 fp16; do
  if [ $m = none ]; then fl=""; else fl="--deform-dtype $m"; fi
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-nystrom $fl > gpurun_out/r4_bench_$m.log 2>&1 || { echo "bench $m rc=$?"; tail -5 gpurun_out/r4_bench_$m.log; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4_bench_$m.log").read().strip().splitlines()[-1])
print("bench", "$m", "ms/step", round(d["ms_per_step"],3), "bags/s", round(d["value"],1), "cpb_bwd ms", round(d["roofline"]["avg_ms"],3), "fwd ms", round(d["roofline_fwd"]["avg_ms"],3))
PY
done
exit $rc
