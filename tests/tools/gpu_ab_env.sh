#!/bin/bash
# A/B of the headline step under one environment switch: gpu_ab_env.sh VAR v0 v1 [repeats]; prints bags/s, ms/step and the event-timed kernels
var=$1; a=$2; b=$3; rep=${4:-2}
mkdir -p gpurun_out/r5
for i in $(seq $rep); do for v in $a $b; do
  env $var=$v timeout -k 10 200 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-nystrom --no-traffic --no-deform16 --no-dp-overhead > gpurun_out/r5/ab.json 2> gpurun_out/r5/ab.err || { tail -5 gpurun_out/r5/ab.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("gpurun_out/r5/ab.json").read().strip().splitlines()[-1])
print("$var=$v", round(d["value"], 1), "bags/s", round(d["ms_per_step"], 3), "ms", {k: round(v["avg_ms"], 3) for k, v in d["kernel_events"].items()})
PY
done; done
