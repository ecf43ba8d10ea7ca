#!/bin/bash
# trip 10: re-run of the tests fixed after trip 9, then the profiler passes of the default command (kernel stats + PMC) for profiles/r04_*
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_data_parallel.py tests/test_gpu_attn16.py tests/test_gpu_deform16.py tests/test_bag_store.py tests/test_gpu_trainstep.py -q -m gpu > gpurun_out/r4_pytest_c.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|FAILED|AssertionError|Error" gpurun_out/r4_pytest_c.log | cut -c1-300 | tail -8
cp gpurun_out/parity_report.tsv gpurun_out/r4_parity_report_c.tsv 2>/dev/null
bash tests/gpu_bench_prof.sh > gpurun_out/r4_bench_prof.log 2>&1; echo "bench prof rc=$?"; tail -5 gpurun_out/r4_bench_prof.log | cut -c1-300
timeout -k 10 300 python tests/bench_modules.py > gpurun_out/r4_modules_bench.txt 2>&1; echo "modules rc=$?"; grep -v amdgpu gpurun_out/r4_modules_bench.txt | tail -12
