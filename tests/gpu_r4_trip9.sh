#!/bin/bash
# trip 9: tests touched since the full-suite run, Nystrom legs with the two-block forward + PMC of the bf16 leg, the default bench (the driver's command)
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 900 python -m pytest tests/test_gpu_data_parallel.py tests/test_gpu_attn16.py tests/test_gpu_deform16.py tests/test_gpu_configs.py -q -m gpu > gpurun_out/r4_pytest_b.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|FAILED|AssertionError|Error" gpurun_out/r4_pytest_b.log | cut -c1-300 | tail -8
cp gpurun_out/parity_report.tsv gpurun_out/r4_parity_report_b.tsv 2>/dev/null
for qb in 0 1; do
SMML_QB=$qb timeout -k 10 300 python - > gpurun_out/r4_nystrom_legs_qb$qb.txt 2>&1 <<'PY'
import importlib, os, sys, torch
sys.path.insert(0, ".")
import bench
pkg = importlib.import_module(bench.PKG)
pkg.lib().smml_attn16_set_query_blocks(int(os.environ["SMML_QB"]))
dev = torch.device("cuda:0")
for B, n, dt in ((8, 4096, torch.bfloat16), (4, 10000, torch.bfloat16), (4, 10000, torch.float16), (1, 50000, torch.float16)):
    r = bench.nystrom_leg(pkg, dev, B, n, dt)
    print(f"query blocks {os.environ['SMML_QB']}: {r['workload']:70s} {r['ms_per_step']:.3f} ms  {r['algorithmic_TFLOPs']:.1f} TF  frac {r['frac']:.4f}")
PY
grep "query blocks" gpurun_out/r4_nystrom_legs_qb$qb.txt
done
bash tests/gpu_nystrom_pmc.sh > gpurun_out/r4_nystrom_pmc.log 2>&1; echo "nystrom pmc rc=$?"
A="tests/bench_nystrom.py --n 10000 --bags 4 --dtype bfloat16 --steps 3"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/nprof -- python $A > gpurun_out/r4_nprof.log 2>&1; echo "nystrom kernel stats rc=$?"
timeout -k 10 900 python bench.py > gpurun_out/r4_bench_default.log 2>&1; echo "default bench rc=$?"; tail -c 2500 gpurun_out/r4_bench_default.log
