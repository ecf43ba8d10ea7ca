#!/bin/bash
# trip 13: A/B of the late layer-1 gate (sign-mask AND, second layer-1 MFMA) in both position-bias backward kernels
set -u
mkdir -p gpurun_out
V=$PWD/subspace-multimodal-learning_amd/lib/variants
for name in base gl base gl; do
  if [ "$name" = base ]; then unset SMML_LIB; else export SMML_LIB=$V/$name.so; fi
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-nystrom > gpurun_out/r4_gl_$name.log 2>&1 || { echo "bench $name rc=$?"; tail -3 gpurun_out/r4_gl_$name.log; continue; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4_gl_$name.log").read().strip().splitlines()[-1])
print("$name fp32 ms/step", round(d["ms_per_step"],3), "bags/s", round(d["value"],1), "cpb", round(d["roofline"]["avg_ms"],3), "| deform16 ms", round(d["deform16"]["ms_per_step"],3), "bags/s", round(d["deform16"]["bags_per_s"],1), "cpb16", round(d["deform16"]["roofline"]["avg_ms"],3))
PY
done
export SMML_LIB=$V/gl.so
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_deform16.py -q -m gpu -k "fused_core or core16 or deform2d or masks" > gpurun_out/r4_gl_tests.log 2>&1
echo "gl tests rc=$?"; grep -E "passed|failed|FAILED" gpurun_out/r4_gl_tests.log | tail -3
