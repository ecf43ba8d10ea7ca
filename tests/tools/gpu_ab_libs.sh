#!/bin/bash
# headline step with each of the given builds of the library (SMML_LIB; lib/variants/<name>.so from tests/tools/build_variants.py, "default" = lib/libsmml_hip.so)
# usage: gpu_ab_libs.sh name [name ...]
mkdir -p gpurun_out/r5
for name in "$@"; do
  lib=subspace-multimodal-learning_amd/lib/variants/$name.so
  [ "$name" = default ] && lib=subspace-multimodal-learning_amd/lib/libsmml_hip.so
  SMML_LIB=$lib timeout -k 10 200 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-nystrom --no-traffic --no-deform16 --no-dp-overhead > gpurun_out/r5/ab.json 2> gpurun_out/r5/ab.err || { tail -5 gpurun_out/r5/ab.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("gpurun_out/r5/ab.json").read().strip().splitlines()[-1])
print("$name", round(d["value"], 1), "bags/s", round(d["ms_per_step"], 3), "ms", {k: round(v["avg_ms"], 3) for k, v in d["kernel_events"].items()})
PY
done
