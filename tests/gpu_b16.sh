#!/bin/bash
# bf16-storage pipeline: tests + GEMM timing (+ build variants in VARIANTS) + the Nystrom legs (NYS=1)
set -u
mkdir -p gpurun_out
P=subspace-multimodal-learning_amd/lib/variants
python -m pytest tests/test_gpu_gemm_b16.py -m gpu -q -x ${PYTEST_K:+-k "$PYTEST_K"} 2>&1 | grep -v amdgpu.ids | tail -5 &&
timeout -k 10 200 python tests/bench_gemm_b16.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gemm_b16_times.txt &&
for v in ${VARIANTS:-}; do echo "== $v"; SMML_LIB=$P/$v.so timeout -k 10 200 python tests/bench_gemm_b16.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/gemm_b16_times.txt || exit 1; done &&
if [ -n "${NYS:-}" ]; then
python -m pytest tests/test_gpu_attn16.py -m gpu -q -x -k "nystrom" 2>&1 | grep -v amdgpu.ids | tail -3 &&
timeout -k 10 200 python tests/bench_nystrom.py --n 10000 --bags 4 --dtype bfloat16 --steps 20 2>&1 | tail -1 | cut -c1-250 &&
SMML_NYSTROM_B16=0 timeout -k 10 200 python tests/bench_nystrom.py --n 10000 --bags 4 --dtype bfloat16 --steps 20 2>&1 | tail -1 | cut -c1-250
fi
