#!/bin/bash
# round 5: everything profiles/r05_* is made from - default bench line, kernel trace of the same command, PMC passes, HBM traffic
set -u
mkdir -p gpurun_out/r5
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 900 python bench.py > gpurun_out/r5/bench_b8.json 2> gpurun_out/r5/bench_b8.err || { echo "bench failed"; tail -5 gpurun_out/r5/bench_b8.err; exit 1; }
rm -rf gpurun_out/r5/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5/prof -- python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-nystrom --no-traffic --no-deform16 --no-dp-overhead > gpurun_out/r5/prof.log 2>&1
echo "prof rc=$?"
bash tests/tools/gpu_r5_pmc.sh
