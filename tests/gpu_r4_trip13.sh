#!/bin/bash
# trip 13: table mode - parity tests, core timing, default bench (with the deform16 and deform16_table legs)
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_deform_table.py -q -m gpu -s > gpurun_out/r4_table_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|FAILED|AssertionError|Error|worst|deform table" gpurun_out/r4_table_pytest.log | cut -c1-900 | tail -12
cp gpurun_out/parity_report.tsv gpurun_out/r4_parity_report_table.tsv 2>/dev/null
timeout -k 10 300 python tests/bench_deform_table.py > gpurun_out/r4_table_core.txt 2>&1
echo "core rc=$?"; tail -7 gpurun_out/r4_table_core.txt
timeout -k 10 600 python bench.py --no-nystrom --no-cpu-baseline --no-traffic > gpurun_out/r4_bench_table.json 2> gpurun_out/r4_bench_table.err
echo "bench rc=$?"; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_bench_table.json").read().strip().splitlines()[-1])
print("fp32-grade", d["value"], d["ms_per_step"])
for k in ("deform16", "deform16_table"):
    x = d.get(k, {})
    print(k, x.get("bags_per_s"), x.get("ms_per_step"), x.get("speedup_vs_fp32_line"), x.get("error"), {a: x[a]["avg_ms"] for a in ("deform_table_fwd", "cpb_table_bwd") if a in x})
PY
