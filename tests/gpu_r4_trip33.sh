#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_nystrom.py tests/test_gpu_attn16.py -q -m gpu > gpurun_out/r4_nys_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|FAILED|AssertionError|Error" gpurun_out/r4_nys_pytest.log | cut -c1-300 | tail -6
timeout -k 10 300 python - > gpurun_out/r4_nystrom_legs2.txt 2>&1 <<'PY'
import importlib, sys, torch
sys.path.insert(0, ".")
import bench
pkg = importlib.import_module(bench.PKG)
dev = torch.device("cuda:0")
for B, n, dt in ((8, 4096, torch.bfloat16), (4, 10000, torch.bfloat16), (4, 10000, torch.float32), (4, 10000, torch.float16), (1, 50000, torch.float16)):
    r = bench.nystrom_leg(pkg, dev, B, n, dt)
    print(f"{r['workload']:70s} {r['ms_per_step']:.3f} ms  {r['algorithmic_TFLOPs']:.1f} TF  frac {r['frac']:.4f}")
PY
grep -v amdgpu gpurun_out/r4_nystrom_legs2.txt | tail -6
