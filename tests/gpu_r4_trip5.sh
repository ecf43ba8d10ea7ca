#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_deform16.py -q -s -m gpu > gpurun_out/r4_deform16_tests.log 2>&1
echo "tests rc=$?"; grep -E "worst rel|passed|failed|FAILED|AssertionError" gpurun_out/r4_deform16_tests.log | cut -c1-500
cp gpurun_out/parity_report.tsv gpurun_out/r4_deform16_parity.tsv 2>/dev/null
export SMML_LIB=$PWD/subspace-multimodal-learning_amd/lib/variants/d2p.so
timeout -k 10 420 python tests/diag_r4_dvs_split.py tumor 100 > gpurun_out/r4_dvs_split_d2p.log 2>&1
echo "dvs split d2p rc=$?"; grep -v amdgpu.ids gpurun_out/r4_dvs_split_d2p.log | cut -c1-200 | sed -n 2,14p; grep -v amdgpu.ids gpurun_out/r4_dvs_split_d2p.log | tail -9 | cut -c1-200
for m in base d2p; do
  if [ $m = base ]; then unset SMML_LIB; else export SMML_LIB=$PWD/subspace-multimodal-learning_amd/lib/variants/d2p.so; fi
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-nystrom --no-deform16 > gpurun_out/r4_bench_$m.log 2>&1 || { echo "bench $m rc=$?"; tail -5 gpurun_out/r4_bench_$m.log; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4_bench_$m.log").read().strip().splitlines()[-1])
print("bench", "$m", "ms/step", round(d["ms_per_step"],3), "bags/s", round(d["value"],1))
PY
done
unset SMML_LIB
for m in bf16 fp16; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-nystrom --no-deform16 --deform-dtype $m > gpurun_out/r4_bench_$m.log 2>&1 || { echo "bench $m rc=$?"; tail -5 gpurun_out/r4_bench_$m.log; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4_bench_$m.log").read().strip().splitlines()[-1])
print("bench", "$m", "ms/step", round(d["ms_per_step"],3), "bags/s", round(d["value"],1), "cpb_bwd ms", round(d["roofline"]["avg_ms"],3), "fwd ms", round(d["roofline_fwd"]["avg_ms"],3))
PY
done
