#!/bin/bash
# Nystrom block, final measurements of the round: tests, timings (eager / hipGraph), kernel trace, PMC passes
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests/test_gpu_gemm_b16.py tests/test_gpu_attn16.py tests/test_gpu_nystrom.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -2 &&
python tests/bench_gemm_b16.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gemm_b16_times.txt &&
python tests/bench_pinv_chain.py 2>&1 | grep "chain form" | tee gpurun_out/pinv_chain_times.txt &&
for args in "--n 10000 --bags 4 --dtype bfloat16" "--n 10000 --bags 4 --dtype bfloat16 --graph" "--n 4096 --bags 8 --dtype bfloat16" "--n 10000 --bags 4 --dtype float32" "--n 10000 --bags 4 --dtype float16" "--n 50000 --bags 1 --dtype float16"; do
  timeout -k 10 200 python tests/bench_nystrom.py $args --steps 20 2>&1 | tail -1 | tee -a gpurun_out/nystrom_legs.txt || exit 1
done &&
rm -rf gpurun_out/prof_nys16 gpurun_out/npmc1 gpurun_out/npmc2 &&
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_nys16 -- python tests/bench_nystrom.py --n 10000 --bags 4 --dtype bfloat16 --steps 8 > gpurun_out/prof_nys16.log 2>&1 &&
f=$(find gpurun_out/prof_nys16 -name "*kernel_stats.csv" | head -1) && cp "$f" gpurun_out/nys16_kernel_stats.csv &&
bash tests/gpu_nystrom_pmc.sh
