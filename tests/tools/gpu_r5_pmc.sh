#!/bin/bash
# round 5: SQ / TCC counters of the region kernels (separate --pmc passes, kernel trace only), summarised by tests/summarize_r5_pmc.py
set -u
mkdir -p gpurun_out/r5
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
A="--steps 3 --warmup 2 --no-cpu-baseline --no-nystrom --no-traffic --no-deform16 --no-dp-overhead ${BENCH_ARGS:-}"
run() { # tag counters...
  local tag=$1; shift
  rm -rf gpurun_out/r5/pmc_$tag
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/r5/pmc_$tag -- python bench.py $A > gpurun_out/r5/pmc_$tag.log 2>&1
  echo "pmc_$tag rc=$?"
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE &&
run b SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS &&
run c SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVES TCC_HIT_sum TCC_MISS_sum &&
if [ "${HBM:-1}" = "1" ]; then run f FETCH_SIZE && run w WRITE_SIZE; fi
python tests/tools/summarize_r5_pmc.py ${PMC_TAG:-r05}
