#!/bin/bash
# quick loop: selected GPU tests (PYTEST_K), two headline bench runs, one kernel trace (-> gpurun_out/${TAG}_kernel_stats.csv)
set -u
mkdir -p gpurun_out
TAG=${TAG:-quick}
if [ -n "${PYTEST_K:-}" ]; then
  python -m pytest tests -m gpu -q -x -k "$PYTEST_K" 2>&1 | grep -v amdgpu.ids | tail -3
fi
for i in 1 2; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nystrom --no-traffic 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms/step', round(d['ms_per_step'],3), 'bags/s', round(d['value'],1), 'cpb_bwd', round(d['roofline']['avg_ms'],3), 'fwd', round(d['roofline_fwd']['avg_ms'],3))"
done
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/prof_$TAG
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-nystrom --no-traffic > gpurun_out/prof_$TAG.log 2>&1
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${TAG}_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/${TAG}_kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)/4e6
print("kernel time per step %.3f ms" % tot)
for r in rows[:8]:
    print("  %-70s %8.3f ms/step  (%d calls)" % (r['Name'][:70], float(r['TotalDurationNs'])/4e6, int(r['Calls'])))
PY
