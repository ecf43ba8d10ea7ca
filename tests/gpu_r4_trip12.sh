#!/bin/bash
# trip 12: two-rank rehearsal of the N > 1 bench path on one GPU (gloo, both ranks on cuda:0 - NOT a scaling figure), then the default bench
set -u
mkdir -p gpurun_out
SMML_BENCH_ONE_DEVICE=1 SMML_DIST_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r4_bench_n2_gloo.log 2>&1
echo "n2 rehearsal rc=$?"; tail -c 1500 gpurun_out/r4_bench_n2_gloo.log
timeout -k 10 900 python bench.py > gpurun_out/r4_bench_default2.log 2>&1; echo "default bench rc=$?"; tail -c 600 gpurun_out/r4_bench_default2.log
