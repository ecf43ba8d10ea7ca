#!/bin/bash
# trip 16: table-forward variant of the 16-bit mode (recomputing backward): parity tests, bench legs
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_deform16.py -q -m gpu -s > gpurun_out/r4_tabfwd_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|FAILED|AssertionError|Error|worst" gpurun_out/r4_tabfwd_pytest.log | cut -c1-700 | tail -14
timeout -k 10 600 python bench.py --no-nystrom --no-cpu-baseline --no-traffic > gpurun_out/r4_bench_tabfwd.json 2> gpurun_out/r4_bench_tabfwd.err
echo "bench rc=$?"; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_bench_tabfwd.json").read().strip().splitlines()[-1])
print("fp32-grade", d["value"], d["ms_per_step"])
for k in ("deform16", "deform16_tabfwd", "deform16_table"):
    x = d.get(k, {})
    print(k, x.get("bags_per_s"), x.get("ms_per_step"), x.get("speedup_vs_fp32_line"), x.get("error"), {a: round(x[a]["avg_ms"], 3) for a in ("deform_table_fwd", "cpb_table_bwd", "roofline") if a in x})
PY
