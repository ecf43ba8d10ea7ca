#!/bin/bash
# bench.py (default run, the driver's command) + rocprofv3 kernel trace + PMC passes (instruction mix, HBM bytes) of the same command.
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
run() { local name=$1 t=$2; shift 2; echo "=== $name"; timeout -k 10 "$t" "$@" > "gpurun_out/$name.log" 2>&1; local rc=$?; echo "rc=$rc"; tail -n 3 "gpurun_out/$name.log" | cut -c1-1500
  if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "!!! $name died (rc $rc): stopping"; exit $rc; fi; return $rc; }
run smoke 300 python -c "import __graft_entry__ as g; g.smoke()" || exit 1
run bench 900 python bench.py || exit 1
A="--steps 3 --warmup 1 --no-cpu-baseline --no-nystrom --no-traffic"
run rocprof 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py $A || exit 1
find gpurun_out/prof -name "*kernel_stats.csv" | head -1 | xargs -r -I{} sh -c 'echo "--- {}"; head -16 {} | cut -c1-200'
run pmc1 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc1 -- python bench.py $A || exit 1
run pmc2 500 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d gpurun_out/pmc2 -- python bench.py $A || exit 1
run pmc3 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc3 -- python bench.py $A || exit 1
run pmc4 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc4 -- python bench.py $A || exit 1
echo done
