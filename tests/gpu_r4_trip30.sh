#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_data_parallel.py -q -m gpu > gpurun_out/r4_rccl1_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|FAILED|AssertionError|Error|skipped" gpurun_out/r4_rccl1_pytest.log | cut -c1-300 | tail -12
