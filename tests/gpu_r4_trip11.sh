#!/bin/bash
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/prof_table
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_table -- python3 tests/bench_deform_table.py > gpurun_out/r4_table_prof.log 2>&1
echo "rc=$?"
f=$(find gpurun_out/prof_table -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:16]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]) / 1e3:9.1f} us  total {float(r["TotalDurationNs"]) / 1e6:8.2f} ms')
PY
