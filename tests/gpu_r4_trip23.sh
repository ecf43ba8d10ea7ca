#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 600 python tests/diag_table_training.py 60 > gpurun_out/r4_table_training.txt 2>&1
echo "rc=$?"; grep -v amdgpu gpurun_out/r4_table_training.txt | tail -14
