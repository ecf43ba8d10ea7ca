#!/bin/bash
# trip 22: full GPU suite + the driver's default bench command
set -u
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -q -m gpu > gpurun_out/r4_pytest_full4.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|FAILED|AssertionError|Error" gpurun_out/r4_pytest_full4.log | cut -c1-300 | tail -8
cp gpurun_out/parity_report.tsv gpurun_out/r4_parity_report_full4.tsv 2>/dev/null
bash tests/gpu_r4_trip17.sh
