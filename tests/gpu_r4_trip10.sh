#!/bin/bash
# trip 10: table mode of the 16-bit core - parity tests and core timing
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_deform_table.py -q -m gpu -s > gpurun_out/r4_table_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|FAILED|AssertionError|Error|worst" gpurun_out/r4_table_pytest.log | cut -c1-600 | tail -12
timeout -k 10 300 python tests/bench_deform_table.py > gpurun_out/r4_table_core.txt 2>&1
echo "bench rc=$?"; cat gpurun_out/r4_table_core.txt | tail -8
