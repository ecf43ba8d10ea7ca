#!/bin/bash
# trip 6: deform16 tests (projection scope fixed), Nystrom 16-bit tests + legs after the fp16-mode change, plateau diag over seeds, PMC of the bf16 step
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 900 python -m pytest tests/test_gpu_deform16.py tests/test_gpu_attn16.py -q -m gpu > gpurun_out/r4_tests16.log 2>&1
echo "tests rc=$?"; grep -E "passed|failed|FAILED|AssertionError" gpurun_out/r4_tests16.log | cut -c1-400
cp gpurun_out/parity_report.tsv gpurun_out/r4_tests16_parity.tsv 2>/dev/null
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -q -m gpu -k "cfg5 or cfg2" > gpurun_out/r4_tests_cfg5.log 2>&1
echo "cfg2/5 rc=$?"; grep -E "passed|failed|FAILED|AssertionError" gpurun_out/r4_tests_cfg5.log | cut -c1-400
timeout -k 10 300 python - > gpurun_out/r4_nystrom_legs.txt 2>&1 <<'PY'
import importlib, json, sys, torch
sys.path.insert(0, ".")
import bench
pkg = importlib.import_module(bench.PKG)
dev = torch.device("cuda:0")
for B, n, dt in ((8, 4096, torch.bfloat16), (4, 10000, torch.bfloat16), (4, 10000, torch.float16), (1, 50000, torch.float16), (4, 10000, torch.float32)):
    r = bench.nystrom_leg(pkg, dev, B, n, dt)
    print(f"{r['workload']:70s} {r['ms_per_step']:.3f} ms  {r['algorithmic_TFLOPs']:.1f} TF  frac {r['frac']:.4f} of {r['peak_TFLOPs']}")
PY
cat gpurun_out/r4_nystrom_legs.txt | grep -v amdgpu
for seed in 1 2 3; do
  timeout -k 10 300 python tests/diag_r4_dvs_split.py tumor 100 $seed > gpurun_out/r4_dvs_split_seed$seed.log 2>&1
  echo "seed $seed rc=$?"; grep -E "HIP total|fp32-oracle total|\(C\)|oracle's dk, exact dv" gpurun_out/r4_dvs_split_seed$seed.log | cut -c1-120
done
A="--steps 3 --warmup 1 --no-cpu-baseline --no-nystrom --no-traffic --no-deform16 --deform-dtype bf16"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof16 -- python bench.py $A > gpurun_out/r4_prof16.log 2>&1; echo "prof rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc16_1 -- python bench.py $A > gpurun_out/r4_pmc16_1.log 2>&1; echo "pmc1 rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d gpurun_out/pmc16_2 -- python bench.py $A > gpurun_out/r4_pmc16_2.log 2>&1; echo "pmc2 rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc16_3 -- python bench.py $A > gpurun_out/r4_pmc16_3.log 2>&1; echo "pmc3 rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc16_4 -- python bench.py $A > gpurun_out/r4_pmc16_4.log 2>&1; echo "pmc4 rc=$?"
echo done
