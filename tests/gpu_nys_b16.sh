#!/bin/bash
# Nystrom block, bf16 bags: tests, eager and hipGraph-replay timings with and without the bf16-storage path, kernel trace of the new path
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python tests/bench_pinv_chain.py 2>&1 | grep "chain form"; python -m pytest tests/test_gpu_gemm_b16.py tests/test_gpu_attn16.py tests/test_gpu_nystrom.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -3 &&
for g in "" "--graph"; do for b in 2; do for cf in 2 1; do
  echo "B16=$b CHAIN=$cf $g"; SMML_CHAIN_FAST=$cf SMML_NYSTROM_B16=$b timeout -k 10 200 python tests/bench_nystrom.py --n 10000 --bags 4 --dtype bfloat16 --steps 20 $g 2>&1 | tail -1 | cut -c80-300 || exit 1
done; done; done &&
rm -rf gpurun_out/prof_nys16 &&
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_nys16 -- python tests/bench_nystrom.py --n 10000 --bags 4 --dtype bfloat16 --steps 8 > gpurun_out/prof_nys16.log 2>&1
echo "rocprof rc=$?"
f=$(find gpurun_out/prof_nys16 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/nys16_kernel_stats.csv; head -30 "$f" | cut -c1-120,200-260
