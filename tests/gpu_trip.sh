#!/bin/bash
# One GPU-box trip: parity tests, smoke, a short bench, and a rocprofv3 kernel-trace of the same bench.
# Steps after a timeout or a death by signal (exit 124 or >= 128) are skipped: a hung or faulted GPU must not be hit again.
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
run() {  # name, timeout, cmd...
  local name=$1 t=$2; shift 2
  echo "=== $name"; timeout -k 10 "$t" "$@" > "gpurun_out/$name.log" 2>&1; local rc=$?
  echo "rc=$rc"; tail -n "${TAILN:-25}" "gpurun_out/$name.log"
  # a timeout (124) or any death by signal (>= 128: 134 abort, 137 kill, 139 segfault after a GPU memory fault) leaves the GPU in an
  # unknown state: nothing else is started on it, the tail of the log above is the evidence
  if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "!!! $name timed out or died by signal (rc $rc): stopping"; exit $rc; fi
  return $rc
}
run pytest_gpu 900 python -m pytest tests -m gpu -q -x ${PYTEST_ARGS:-}
PYRC=$?
run smoke 300 python -c "import __graft_entry__ as g; g.smoke()"
run bench 600 python bench.py --steps ${STEPS:-5} --warmup 2 ${BENCH_ARGS:-}
if [ "${PROFILE:-1}" = "1" ]; then
  run rocprof 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-}
  find gpurun_out/prof -name "*kernel_stats.csv" | head -1 | xargs -r -I{} sh -c 'echo "--- {}"; head -25 {}'
fi
exit $PYRC
