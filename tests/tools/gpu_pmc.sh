#!/bin/bash
# PMC passes (separate from --stats runs, as the pool requires): MFMA busy, wave cycles, waits, LDS conflicts.
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
A="--steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-}"
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc1 -- python bench.py $A > gpurun_out/pmc1.log 2>&1
echo "pmc1 rc=$?"; tail -3 gpurun_out/pmc1.log | cut -c1-300
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d gpurun_out/pmc2 -- python bench.py $A > gpurun_out/pmc2.log 2>&1
echo "pmc2 rc=$?"; tail -3 gpurun_out/pmc2.log | cut -c1-300
if [ "${HBM:-0}" = "1" ]; then
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc3 -- python bench.py $A > gpurun_out/pmc3.log 2>&1
echo "pmc3 rc=$?"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc4 -- python bench.py $A > gpurun_out/pmc4.log 2>&1
echo "pmc4 rc=$?"
fi
find gpurun_out/pmc1 gpurun_out/pmc2 -name "*.csv" | head
