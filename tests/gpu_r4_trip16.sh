#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py -q -m gpu -k "cfg4" > gpurun_out/r4_cfg4.log 2>&1
echo "cfg4 rc=$?"; grep -E "passed|failed|FAILED|AssertionError" gpurun_out/r4_cfg4.log | cut -c1-300 | tail -5
grep "nothing imposed" gpurun_out/parity_report.tsv | awk -F'\t' '{print $2, $3, $4, $5}' | sort -k6 -g -r | head -12
cp gpurun_out/parity_report.tsv gpurun_out/r4_parity_cfg4.tsv
