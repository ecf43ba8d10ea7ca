#!/bin/bash
# r03 step A: full GPU suite with the decision-conditioned gate on the default build, then the parked variants under the same gate.
set -u
mkdir -p gpurun_out
PYTEST_ARGS="" SUITE_TIMEOUT=900 TAILN=40 bash tests/gpu_suite.sh; rc=$?
cp gpurun_out/parity_report.tsv gpurun_out/parity_report_default.tsv 2>/dev/null
cp gpurun_out/pytest_gpu.log gpurun_out/pytest_gpu_default.log
if [ $rc -ge 124 ]; then exit $rc; fi
V=subspace-multimodal-learning_amd/lib/variants
for name in s3 s3b2; do
  echo "=== variant $name"
  SMML_LIB=$PWD/$V/$name.so timeout -k 10 700 python -m pytest tests -m gpu -q -k "deform or fused or cfg4 or mil_branch or full_model or relu or consistent or true_1d" > gpurun_out/pytest_gpu_$name.log 2>&1
  r=$?; echo "rc=$r"; grep -v amdgpu.ids gpurun_out/pytest_gpu_$name.log | tail -n 15
  cp gpurun_out/parity_report.tsv gpurun_out/parity_report_$name.tsv 2>/dev/null
  if [ $r -ge 124 ]; then exit $r; fi
done
echo done
