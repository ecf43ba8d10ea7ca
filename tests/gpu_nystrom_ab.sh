set -u
for ov in 1 0 1 0; do
  for args in "--n 10000 --bags 4 --dtype bfloat16" "--n 10000 --bags 4 --dtype float32" "--n 4096 --bags 8 --dtype bfloat16"; do
    echo -n "overlap=$ov $args: "; SMML_NYSTROM_OVERLAP=$ov timeout -k 10 300 python tests/bench_nystrom.py $args 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
