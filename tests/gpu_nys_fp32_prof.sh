#!/bin/bash
# kernel trace of the Nystrom block's exact fp32 step (4 x 10 000 x 512)
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/prof_nys32
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_nys32 -- python tests/bench_nystrom.py --n 10000 --bags 4 --dtype float32 --steps 8 > gpurun_out/prof_nys32.log 2>&1
f=$(find gpurun_out/prof_nys32 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/nys32_kernel_stats.csv; tail -1 gpurun_out/prof_nys32.log | cut -c1-200
