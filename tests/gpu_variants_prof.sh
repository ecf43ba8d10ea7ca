set -u
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
V=subspace-multimodal-learning_amd/lib/variants
for name in $VARIANTS; do
  echo "=== $name"
  SMML_LIB=$PWD/$V/$name.so timeout -k 10 300 python tests/diag_gterms.py 2>&1 | grep -v amdgpu | grep "out \|dq \|dk \|dv \|dvs \|dw1 \|dw3 "
  rm -rf gpurun_out/pv_$name
  SMML_LIB=$PWD/$V/$name.so timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pv_$name -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-nystrom > gpurun_out/pv_$name.log 2>&1 || { echo fail; tail -3 gpurun_out/pv_$name.log; exit 1; }
  f=$(find gpurun_out/pv_$name -name "*kernel_stats.csv" | head -1); grep "bwd_dq\|bwd_dkv\|cpb_bwd\|attn_fwd" $f | cut -d, -f1-4 | sed 's/(anonymous namespace):://g' | cut -c1-60,150-
done
