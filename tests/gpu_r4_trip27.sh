#!/bin/bash
# trip 27: soak of the new kernels - more random shapes, other seeds
set -u
mkdir -p gpurun_out
for seed in 7 31; do
SMML_FUZZ_CASES=40 SMML_FUZZ_GRIDS=24 SMML_FUZZ_SEED=$seed timeout -k 10 900 python -m pytest tests/test_gpu_deform_table.py tests/test_gpu_deform16.py -q -m gpu -k "random_shapes or grid_queries" > gpurun_out/r4_soak_$seed.log 2>&1
echo "seed $seed rc=$?"; grep -E "passed|failed|FAILED|AssertionError" gpurun_out/r4_soak_$seed.log | cut -c1-300 | tail -6
done
