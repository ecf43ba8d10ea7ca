#!/bin/bash
# trip 24: hipGraph replay of the table-forward step; two-rank gloo rehearsal of the bench on one device
set -u
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --graph --deform-dtype bf16 --deform-table forward --no-cpu-baseline --no-traffic --no-nystrom --no-deform16 > gpurun_out/r4_graph_tabfwd.log 2>&1
echo "graph rc=$?"; tail -1 gpurun_out/r4_graph_tabfwd.log | cut -c1-260
SMML_BENCH_ONE_DEVICE=1 SMML_DIST_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --bags 4 > gpurun_out/r4_bench_n2_rehearsal.json 2> gpurun_out/r4_bench_n2_rehearsal.err
echo "n2 rc=$?"; tail -1 gpurun_out/r4_bench_n2_rehearsal.json | cut -c1-400
