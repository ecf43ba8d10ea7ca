#!/bin/bash
# Accuracy (tests/diag_gterms.py) and step time (bench.py) of library variants built by tests/build_variants.py.
set -u
mkdir -p gpurun_out
V=subspace-multimodal-learning_amd/lib/variants
for name in ${VARIANTS:-base s5 dfix s5dfix}; do
  echo "=== $name"
  SMML_LIB=$PWD/$V/$name.so timeout -k 10 300 python tests/diag_gterms.py > gpurun_out/var_${name}_diag.log 2>&1 || { echo "diag rc=$?"; tail -5 gpurun_out/var_${name}_diag.log; exit 1; }
  cat gpurun_out/var_${name}_diag.log
  SMML_LIB=$PWD/$V/$name.so timeout -k 10 300 python tests/diag_gterms.py 3000 300 active ${ACTIVE_MODE:-small} > gpurun_out/var_${name}_diag_active.log 2>&1 || { echo "diag rc=$?"; exit 1; }
  cat gpurun_out/var_${name}_diag_active.log
  SMML_LIB=$PWD/$V/$name.so timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > gpurun_out/var_${name}_bench.log 2>&1 || { echo "bench rc=$?"; tail -5 gpurun_out/var_${name}_bench.log; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/var_${name}_bench.log").read().strip().splitlines()[-1])
print("bench", "$name", "ms/step", round(d["ms_per_step"],3), "bags/s", round(d["value"],1), "cpb_bwd ms", round(d["roofline"]["avg_ms"],3), "fwd ms", round(d["roofline_fwd"]["avg_ms"],3))
PY
done
