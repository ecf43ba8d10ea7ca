#!/bin/bash
# trip 2: deform16 parity + step times, then the bisect of the tumor-branch plateau with NO direct vgrid term in the loss (VG_W = 0)
set -u
mkdir -p gpurun_out
bash tests/gpu_r4_deform16.sh
echo "deform16 rc=$?"
V=$PWD/subspace-multimodal-learning_amd/lib/variants
for name in base allp; do
  echo "=== diag VG_W=0 $name"
  if [ "$name" = base ]; then unset SMML_LIB; else export SMML_LIB=$V/$name.so; fi
  timeout -k 10 420 python tests/diag_cfg4_branch.py tumor 100 0 > gpurun_out/r4_diag0_${name}.log 2>&1 || { echo "diag rc=$?"; tail -5 gpurun_out/r4_diag0_${name}.log; exit 1; }
  grep -v amdgpu.ids gpurun_out/r4_diag0_${name}.log | cut -c1-220
done
