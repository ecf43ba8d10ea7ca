#!/bin/bash
# round 5: quick A/B of the headline step, region kernels (default) vs per-pair MLP kernels (SMML_CPB_REGIONS=0), + kernel trace
mkdir -p gpurun_out/r5
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for mode in 1 0; do
  SMML_CPB_REGIONS=$mode timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nystrom --no-traffic --no-deform16 --no-dp-overhead > gpurun_out/r5/quick_regions$mode.json 2> gpurun_out/r5/quick_regions$mode.err || exit 1
  python - <<PY
import json
d = json.loads(open("gpurun_out/r5/quick_regions$mode.json").read().strip().splitlines()[-1])
print("regions=$mode", round(d["value"], 1), "bags/s", round(d["ms_per_step"], 3), "ms", {k: round(v["avg_ms"], 3) for k, v in d["kernel_events"].items()})
PY
done
rm -rf gpurun_out/r5/prof_quick
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5/prof_quick -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-nystrom --no-traffic --no-deform16 --no-dp-overhead > /dev/null 2>&1
f=$(find gpurun_out/r5/prof_quick -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -d, -f1-4 "$f" | sed 's/(anonymous namespace):://g' | cut -c1-140 | head -40
