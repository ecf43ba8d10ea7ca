#!/bin/bash
# trip 15: full GPU suite, hipGraph replay of the table-mode step
set -u
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/r4_pytest_full2.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|FAILED|AssertionError|Error" gpurun_out/r4_pytest_full2.log | cut -c1-300 | tail -8
cp gpurun_out/parity_report.tsv gpurun_out/r4_parity_report_full2.tsv 2>/dev/null
timeout -k 10 300 python bench.py --graph --deform-dtype bf16 --deform-table --no-cpu-baseline --no-traffic --no-nystrom --no-deform16 > gpurun_out/r4_graph_table.log 2>&1
echo "graph rc=$?"; tail -1 gpurun_out/r4_graph_table.log | cut -c1-400
