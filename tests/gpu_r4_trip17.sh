#!/bin/bash
# trip 17: the driver's default command, timed
set -u
mkdir -p gpurun_out
t0=$(date +%s)
timeout -k 10 900 python bench.py > gpurun_out/r4_bench_default2.json 2> gpurun_out/r4_bench_default2.err
echo "bench rc=$? seconds=$(( $(date +%s) - t0 ))"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_bench_default2.json").read().strip().splitlines()[-1])
print("value", d["value"], d["ms_per_step"], "roofline frac", d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"))
for k in ("deform16", "deform16_tabfwd", "deform16_table"):
    x = d.get(k, {})
    print(k, x.get("bags_per_s"), x.get("ms_per_step"), x.get("speedup_vs_fp32_line"), x.get("error"))
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("sample", "")[:80])
print("nystrom", [(l.get("workload", "")[:40], round(l.get("ms_per_step", 0), 2)) for l in d.get("nystrom", {}).get("legs", [])] if isinstance(d.get("nystrom"), dict) else d.get("nystrom"))
PY
