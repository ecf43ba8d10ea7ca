#!/bin/bash
# trip 15: A/B of two keys per trip in the fp32-grade forward
set -u
mkdir -p gpurun_out
V=$PWD/subspace-multimodal-learning_amd/lib/variants
for name in base fpair base fpair; do
  if [ "$name" = base ]; then unset SMML_LIB; else export SMML_LIB=$V/$name.so; fi
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-nystrom --no-deform16 > gpurun_out/r4_fp_$name.log 2>&1 || { echo "bench $name rc=$?"; tail -3 gpurun_out/r4_fp_$name.log; continue; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4_fp_$name.log").read().strip().splitlines()[-1])
print("$name fp32 ms/step", round(d["ms_per_step"],3), "bags/s", round(d["value"],1), "cpb", round(d["roofline"]["avg_ms"],3), "fwd", round(d["roofline_fwd"]["avg_ms"],3))
PY
done
export SMML_LIB=$V/fpair.so
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "fused_core or deform2d or masks" > gpurun_out/r4_fp_tests.log 2>&1
echo "fpair tests rc=$?"; grep -E "passed|failed|FAILED" gpurun_out/r4_fp_tests.log | tail -3
