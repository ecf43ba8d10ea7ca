#!/bin/bash
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/nprof16f -- python tests/bench_nystrom.py --n 50000 --bags 1 --dtype float16 --steps 3 > gpurun_out/r4_nprof16f.log 2>&1; echo "rc=$?"
find gpurun_out/nprof16f -name "*kernel_stats.csv" | head -1 | xargs -r -I{} sh -c 'head -30 {} | cut -c1-170'
