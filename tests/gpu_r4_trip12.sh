#!/bin/bash
set -u
mkdir -p gpurun_out
SMML_TABLE_NOHIST=1 timeout -k 10 300 python tests/bench_deform_table.py > gpurun_out/r4_table_core_nohist.txt 2>&1
echo "bench rc=$?"; tail -6 gpurun_out/r4_table_core_nohist.txt
