#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_deform_table.py -q -m gpu > gpurun_out/r4_table_pytest3.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|FAILED|AssertionError|Error" gpurun_out/r4_table_pytest3.log | cut -c1-300 | tail -4
timeout -k 10 300 python tests/bench_deform_table.py > gpurun_out/r4_table_core3.txt 2>&1; tail -6 gpurun_out/r4_table_core3.txt
timeout -k 10 600 python bench.py --no-nystrom --no-cpu-baseline --no-traffic > gpurun_out/r4_bench_table3.json 2> gpurun_out/r4_bench_table3.err
echo "bench rc=$?"; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_bench_table3.json").read().strip().splitlines()[-1])
print("fp32-grade", d["value"], d["ms_per_step"])
for k in ("deform16", "deform16_tabfwd", "deform16_table"):
    x = d.get(k, {})
    print(k, x.get("bags_per_s"), x.get("ms_per_step"), x.get("speedup_vs_fp32_line"), x.get("error"), {a: round(x[a]["avg_ms"], 3) for a in ("deform_table_fwd", "cpb_table_bwd") if a in x})
PY
