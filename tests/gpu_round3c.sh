#!/bin/bash
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
export SMML_GEMM_MODE=2
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03c -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-nystrom --no-traffic > gpurun_out/prof_r03c.log 2>&1
f=$(find gpurun_out/prof_r03c -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r03c_kernel_stats.csv; grep -i gemm "$f" | cut -c1-140
timeout -k 10 900 python -m pytest tests -m gpu -q -k "not cfg4_full_fusion_10000x512 and not cfg5" > gpurun_out/pytest_gm2.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids gpurun_out/pytest_gm2.log | tail -6
