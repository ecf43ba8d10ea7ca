#!/bin/bash
# builds one microbench binary per variant into tests/microbench/bin/ (git-ignored, travels with gpurun)
set -e
cd "$(dirname "$0")"
mkdir -p bin && rm -f bin/mb_*
build() { # name, flags...
  local name=$1; shift
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -fno-honor-nans -mno-amdgpu-ieee -fno-slp-vectorize -DVARIANT="\"$name\"" "$@" attn_microbench.hip -o bin/mb_$name &
}
build base
build deltafix -DSMML_DELTA_FIX=1
wait
ls bin
