#!/bin/bash
# final trip of the round: the complete GPU suite on the final code, then the driver's default command with its profiler passes
set -u
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -q -m gpu > gpurun_out/r4_pytest_final.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|FAILED|AssertionError|Error" gpurun_out/r4_pytest_final.log | cut -c1-300 | tail -6
cp gpurun_out/parity_report.tsv gpurun_out/r4_parity_report_final.tsv 2>/dev/null
bash tests/gpu_bench_prof.sh > gpurun_out/r4_bench_prof_final.log 2>&1; echo "bench prof rc=$?"
tail -c 400 gpurun_out/bench.log
