#!/bin/bash
# few-keys forward: tests + kernel time (rocprofv3 kernel trace of the Nystrom 16-bit step)
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests/test_gpu_attn16.py -m gpu -q -x -k "fewkeys or merged or nystrom_16bit" 2>&1 | grep -v amdgpu.ids | tail -2
rm -rf gpurun_out/prof_nys16
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_nys16 -- python tests/bench_nystrom.py --n 10000 --bags 4 --dtype bfloat16 --steps 5 > gpurun_out/prof_nys16.log 2>&1
f=$(find gpurun_out/prof_nys16 -name "*kernel_stats.csv" | head -1); grep -E "attn16" "$f" | cut -c1-60,100-200
tail -1 gpurun_out/prof_nys16.log | cut -c1-200
