#!/bin/bash
# trip 3: deform16 parity (fp16 scores), kernel-level profile of the bf16 step, d vs split diagnostic
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 900 python -m pytest tests/test_gpu_deform16.py -q -s -m gpu -k "not cfg4" > gpurun_out/r4_deform16_tests.log 2>&1
echo "tests rc=$?"; grep -E "worst|passed|failed|FAILED|Error" gpurun_out/r4_deform16_tests.log | cut -c1-400
A="--steps 3 --warmup 1 --no-cpu-baseline --no-nystrom --no-traffic --no-deform16"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof16 -- python bench.py $A --deform-dtype bf16 > gpurun_out/r4_prof16.log 2>&1
echo "rocprof rc=$?"
find gpurun_out/prof16 -name "*kernel_stats.csv" | head -1 | xargs -r -I{} sh -c 'head -40 {} | cut -c1-160'
timeout -k 10 420 python tests/diag_r4_dvs_split.py tumor 100 > gpurun_out/r4_dvs_split.log 2>&1
echo "dvs split rc=$?"; grep -v amdgpu.ids gpurun_out/r4_dvs_split.log | cut -c1-200
