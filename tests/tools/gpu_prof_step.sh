#!/bin/bash
# kernel trace of the headline step (rocprofv3 --kernel-trace --stats) next to the same run's event-timed kernels without the profiler;
# usage: gpu_prof_step.sh [steps] [warmup]  ->  gpurun_out/r5/prof_step/
steps=${1:-10}; warm=${2:-5}
mkdir -p gpurun_out/r5
args="--steps $steps --warmup $warm --no-cpu-baseline --no-nystrom --no-traffic --no-deform16 --no-dp-overhead"
timeout -k 10 300 python bench.py $args > gpurun_out/r5/prof_step_plain.json 2> gpurun_out/r5/prof_step_plain.err || exit 1
python - <<PY
import json
d = json.loads(open("gpurun_out/r5/prof_step_plain.json").read().strip().splitlines()[-1])
print("no profiler:", round(d["value"], 1), "bags/s", round(d["ms_per_step"], 3), "ms", {k: round(v["avg_ms"], 3) for k, v in d["kernel_events"].items()})
PY
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/r5/prof_step
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5/prof_step -- python bench.py $args > gpurun_out/r5/prof_step.json 2> gpurun_out/r5/prof_step.err || exit 1
python - <<PY
import csv, glob, json
d = json.loads(open("gpurun_out/r5/prof_step.json").read().strip().splitlines()[-1])
print("under rocprofv3:", round(d["value"], 1), "bags/s", round(d["ms_per_step"], 3), "ms", {k: round(v["avg_ms"], 3) for k, v in d["kernel_events"].items()})
f = sorted(glob.glob("gpurun_out/r5/prof_step/*/*kernel_stats.csv"))[-1]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r["Name"].replace("(anonymous namespace)::", "")[:60], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us avg", round(float(r["MinNs"]) / 1e3, 1), "min", round(float(r["MaxNs"]) / 1e3, 1), "max")
PY
