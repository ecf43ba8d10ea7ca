#!/bin/bash
# trip 11: A/B of GradFork on both lines, then the complete GPU suite on the final code
set -u
mkdir -p gpurun_out
for f in 1 0 1 0; do
  SMML_GRAD_FORK=$f timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-nystrom > gpurun_out/r4_fork_$f.log 2>&1 || { echo "bench fork=$f rc=$?"; tail -3 gpurun_out/r4_fork_$f.log; continue; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4_fork_$f.log").read().strip().splitlines()[-1])
print("fork=$f fp32 ms/step", round(d["ms_per_step"],3), "bags/s", round(d["value"],1), "| deform16 ms", round(d["deform16"]["ms_per_step"],3), "bags/s", round(d["deform16"]["bags_per_s"],1))
PY
done
timeout -k 10 1100 python -m pytest tests -q -m gpu > gpurun_out/r4_pytest_full.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|FAILED|AssertionError|Error" gpurun_out/r4_pytest_full.log | cut -c1-300 | tail -8
cp gpurun_out/parity_report.tsv gpurun_out/r4_parity_report_full.tsv 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
