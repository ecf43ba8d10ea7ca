#!/bin/bash
# attention kernels of the bf16 Nystrom step: tests + kernel times from a rocprofv3 trace
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests/test_gpu_attn16.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -2 &&
rm -rf gpurun_out/prof_nys16 &&
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_nys16 -- python tests/bench_nystrom.py --n 10000 --bags 4 --dtype bfloat16 --steps 8 > gpurun_out/prof_nys16.log 2>&1 &&
f=$(find gpurun_out/prof_nys16 -name "*kernel_stats.csv" | head -1) && grep attn16 "$f" | awk -F'","' '{printf "%-100s %8.1f us\n", substr($1,1,100), $4/1000}' &&
for i in 1 2; do python tests/bench_nystrom.py --n 10000 --bags 4 --dtype bfloat16 --steps 30 2>&1 | tail -1 | cut -c80-140; done
