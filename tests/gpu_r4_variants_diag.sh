#!/bin/bash
# which precision knob moves the tumor-branch gradient plateau of the full-size cfg4 case?
# Runs tests/diag_cfg4_branch.py (tumor, 100 x 100) once per library variant; logs under gpurun_out/r4_diag_<name>.log
set -u
mkdir -p gpurun_out
V=$PWD/subspace-multimodal-learning_amd/lib/variants
for name in ${VARIANTS:-base dkv3 dqo3 dfix all3}; do
  echo "=== $name"
  if [ "$name" = base ]; then unset SMML_LIB; else export SMML_LIB=$V/$name.so; fi
  timeout -k 10 420 python tests/diag_cfg4_branch.py tumor 100 > gpurun_out/r4_diag_${name}.log 2>&1 || { echo "diag rc=$?"; tail -5 gpurun_out/r4_diag_${name}.log; exit 1; }
  grep -E "branch|d vs|to_offsets|to_q|omic_net|fusion_layer|norm" gpurun_out/r4_diag_${name}.log | cut -c1-200
done
