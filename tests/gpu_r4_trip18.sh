#!/bin/bash
# trip 18: full GPU suite; kernel trace of the table-forward step
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 1100 python -m pytest tests -q -m gpu > gpurun_out/r4_pytest_full3.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|FAILED|AssertionError|Error" gpurun_out/r4_pytest_full3.log | cut -c1-300 | tail -8
cp gpurun_out/parity_report.tsv gpurun_out/r4_parity_report_full3.tsv 2>/dev/null
A="--steps 3 --warmup 1 --no-cpu-baseline --no-nystrom --no-traffic --no-deform16 --deform-dtype bf16 --deform-table forward"
rm -rf gpurun_out/proftabfwd
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/proftabfwd -- python bench.py $A > gpurun_out/r4_proftabfwd.log 2>&1; echo "prof rc=$?"
