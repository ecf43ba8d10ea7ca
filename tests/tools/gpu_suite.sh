#!/bin/bash
# Full GPU parity suite with the report kept (gpurun_out/parity_report.tsv, gpurun_out/pytest_gpu.log).
set -u
mkdir -p gpurun_out
timeout -k 10 ${SUITE_TIMEOUT:-1100} python -m pytest tests -m gpu -q ${PYTEST_ARGS:--x} ${PYTEST_K:+-k "$PYTEST_K"} --durations=15 > gpurun_out/pytest_gpu.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -v amdgpu.ids gpurun_out/pytest_gpu.log | tail -n ${TAILN:-60}
exit $rc
