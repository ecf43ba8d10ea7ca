#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_deform16.py -q -s -m gpu > gpurun_out/r4_deform16_tests.log 2>&1
echo "tests rc=$?"; grep -E "worst|passed|failed|FAILED|AssertionError" gpurun_out/r4_deform16_tests.log | cut -c1-600
timeout -k 10 420 python tests/diag_r4_dvs_split.py tumor 100 > gpurun_out/r4_dvs_split.log 2>&1
echo "dvs split rc=$?"; grep -v amdgpu.ids gpurun_out/r4_dvs_split.log | tail -12 | cut -c1-200
