#!/bin/bash
# trip 25: PMC passes of the table-forward step
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
A="--steps 3 --warmup 1 --no-cpu-baseline --no-nystrom --no-traffic --no-deform16 --deform-dtype bf16 --deform-table forward"
rm -rf gpurun_out/pmctf_1 gpurun_out/pmctf_2 gpurun_out/pmctf_3 gpurun_out/pmctf_4
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmctf_1 -- python bench.py $A > gpurun_out/r4_pmctf_1.log 2>&1; echo "pmc1 rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d gpurun_out/pmctf_2 -- python bench.py $A > gpurun_out/r4_pmctf_2.log 2>&1; echo "pmc2 rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmctf_3 -- python bench.py $A > gpurun_out/r4_pmctf_3.log 2>&1; echo "pmc3 rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmctf_4 -- python bench.py $A > gpurun_out/r4_pmctf_4.log 2>&1; echo "pmc4 rc=$?"
echo done
