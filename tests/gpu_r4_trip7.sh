#!/bin/bash
# trip 7: A/B of the 16-bit kernels' variants on one box (bench --deform-dtype bf16), dropout tests after the two-decisions-per-hash change
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_deform16.py tests/test_gpu_options.py -q -m gpu -k "dropout or drop or core16 or fused_core or train_mode or graph" > gpurun_out/r4_tests_drop.log 2>&1
echo "dropout tests rc=$?"; grep -E "passed|failed|FAILED|AssertionError" gpurun_out/r4_tests_drop.log | cut -c1-300
V=$PWD/subspace-multimodal-learning_amd/lib/variants
for name in base gm fp2 gmfp2 base; do
  if [ "$name" = base ]; then unset SMML_LIB; else export SMML_LIB=$V/$name.so; fi
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-nystrom --no-deform16 --deform-dtype bf16 > gpurun_out/r4_v_$name.log 2>&1 || { echo "bench $name rc=$?"; tail -3 gpurun_out/r4_v_$name.log; continue; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4_v_$name.log").read().strip().splitlines()[-1])
print("bench16", "$name", "ms/step", round(d["ms_per_step"],3), "bags/s", round(d["value"],1), "cpb16 ms", round(d["roofline"]["avg_ms"],3), "fwd16 ms", round(d["roofline_fwd"]["avg_ms"],3))
PY
done
unset SMML_LIB
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-nystrom --no-deform16 > gpurun_out/r4_v_fp32.log 2>&1
python - <<PY
import json
d=json.loads(open("gpurun_out/r4_v_fp32.log").read().strip().splitlines()[-1])
print("bench fp32", "ms/step", round(d["ms_per_step"],3), "bags/s", round(d["value"],1), "cpb ms", round(d["roofline"]["avg_ms"],3), "fwd ms", round(d["roofline_fwd"]["avg_ms"],3))
PY
