#!/bin/bash
set -u
mkdir -p gpurun_out
for m in fused foreach fused foreach; do
  SMML_ADAM=$m timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-nystrom > gpurun_out/r4_adam_$m.log 2>&1 || { echo "bench adam=$m rc=$?"; tail -3 gpurun_out/r4_adam_$m.log; continue; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4_adam_$m.log").read().strip().splitlines()[-1])
print("adam=$m fp32 ms/step", round(d["ms_per_step"],3), "bags/s", round(d["value"],1), "| deform16 ms", round(d["deform16"]["ms_per_step"],3), "bags/s", round(d["deform16"]["bags_per_s"],1))
PY
done
