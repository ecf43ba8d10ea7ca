#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_deform_table.py -q -m gpu > gpurun_out/r4_table_pytest2.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|FAILED|AssertionError|Error" gpurun_out/r4_table_pytest2.log | cut -c1-300 | tail -6
timeout -k 10 300 python tests/bench_deform_table.py > gpurun_out/r4_table_core2.txt 2>&1
echo "core rc=$?"; tail -6 gpurun_out/r4_table_core2.txt
