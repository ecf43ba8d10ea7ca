#!/bin/bash
cd "$(dirname "$0")"
# MB_REPS rounds over all variants (interleaved: clocks drift by a percent or two between runs - compare minima)
for rep in $(seq 1 ${MB_REPS:-1}); do
  for b in bin/mb_*; do timeout -k 5 120 $b ${MB_ARGS:-4 100} || echo "$b failed rc=$?"; done
done
if [ -n "${MB_PROF:-}" ]; then   # per-kernel times of one variant: MB_PROF=base
  cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
  out=../../gpurun_out/mbprof; rm -rf $out
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- bin/mb_$MB_PROF ${MB_ARGS:-4 100} > /dev/null 2>&1
  find $out -name "*kernel_stats.csv" | xargs -r -I{} sh -c 'cut -d, -f1-4 {} | cut -c1-90,200-'
fi
