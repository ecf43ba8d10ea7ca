#!/bin/bash
# trip 8: A/B of the d p-on-the-matrix-pipe variant, then the complete GPU suite
set -u
mkdir -p gpurun_out
V=$PWD/subspace-multimodal-learning_amd/lib/variants
for name in base dpm base dpm; do
  if [ "$name" = base ]; then unset SMML_LIB; else export SMML_LIB=$V/$name.so; fi
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-nystrom --no-deform16 --deform-dtype bf16 > gpurun_out/r4_v_$name.log 2>&1 || { echo "bench $name rc=$?"; tail -3 gpurun_out/r4_v_$name.log; continue; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4_v_$name.log").read().strip().splitlines()[-1])
print("bench16", "$name", "ms/step", round(d["ms_per_step"],3), "bags/s", round(d["value"],1), "cpb16 ms", round(d["roofline"]["avg_ms"],3), "fwd16 ms", round(d["roofline_fwd"]["avg_ms"],3))
PY
done
unset SMML_LIB
timeout -k 10 1100 python -m pytest tests -q -m gpu -x > gpurun_out/r4_pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|FAILED|AssertionError|Error" gpurun_out/r4_pytest_gpu.log | cut -c1-300 | tail -8
cp gpurun_out/parity_report.tsv gpurun_out/r4_parity_report_a.tsv 2>/dev/null
