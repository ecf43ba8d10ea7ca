#!/bin/bash
# trip 17: fused BatchLoss tail - parity tests that exercise it, then A/B on both bench lines
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_data_parallel.py tests/test_gpu_configs.py -q -m gpu -k "batch_loss or full_model or mil_branch or two_ranks or (cfg4 and 24)" > gpurun_out/r4_bl_tests.log 2>&1
echo "tests rc=$?"; grep -E "passed|failed|FAILED|AssertionError" gpurun_out/r4_bl_tests.log | cut -c1-300 | tail -5
for f in 1 0 1 0; do
  SMML_BATCHLOSS_TAIL=$f timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-nystrom > gpurun_out/r4_bl_$f.log 2>&1 || { echo "bench tail=$f rc=$?"; tail -3 gpurun_out/r4_bl_$f.log; continue; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4_bl_$f.log").read().strip().splitlines()[-1])
print("tail=$f fp32 ms/step", round(d["ms_per_step"],3), "bags/s", round(d["value"],1), "| deform16 ms", round(d["deform16"]["ms_per_step"],3), "bags/s", round(d["deform16"]["bags_per_s"],1))
PY
done
