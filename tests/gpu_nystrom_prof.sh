#!/bin/bash
# Nystrom block: timing (fp32 mode and 16-bit mode) and a rocprofv3 kernel trace of the 16-bit mode at n = 10 000.
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for args in "--n 10000 --bags 4 --dtype float32" "--n 10000 --bags 4 --dtype bfloat16" "--n 10000 --bags 4 --dtype float16" "--n 4096 --bags 8 --dtype bfloat16" "--n 50000 --bags 1 --dtype float16"; do
  timeout -k 10 300 python tests/bench_nystrom.py $args 2>&1 | grep -v amdgpu.ids | tail -1
done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_nys16 -- python tests/bench_nystrom.py --n 10000 --bags 4 --dtype bfloat16 --steps 5 > gpurun_out/prof_nys16.log 2>&1
echo "rocprof rc=$?"
f=$(find gpurun_out/prof_nys16 -name "*kernel_stats.csv" | head -1); echo "--- $f"; head -22 "$f" | cut -c1-170
