#!/bin/bash
# PMC passes (instruction mix) of the Nystrom block's 16-bit step: what bounds the fused attention kernels.
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
A="tests/tools/bench_nystrom.py --n 10000 --bags 4 --dtype bfloat16 --steps 3"
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/npmc1 -- python $A > gpurun_out/npmc1.log 2>&1 || { echo "pmc1 failed"; tail -3 gpurun_out/npmc1.log; exit 1; }
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d gpurun_out/npmc2 -- python $A > gpurun_out/npmc2.log 2>&1 || { echo "pmc2 failed"; tail -3 gpurun_out/npmc2.log; exit 1; }
# HBM traffic, one counter per pass (MI355X_MICROARCH.md, HBM / rocprofv3 section)
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/npmc3 -- python $A > gpurun_out/npmc3.log 2>&1 || { echo "pmc3 failed"; tail -3 gpurun_out/npmc3.log; exit 1; }
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/npmc4 -- python $A > gpurun_out/npmc4.log 2>&1 || { echo "pmc4 failed"; tail -3 gpurun_out/npmc4.log; exit 1; }
echo done
