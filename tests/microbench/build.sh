#!/bin/bash
# builds one microbench binary per variant into gpurun-visible tests/microbench/bin/ (git-ignored via *.o? no: listed in .gitignore)
set -e
cd "$(dirname "$0")"
mkdir -p bin
build() { # name, flags...
  local name=$1; shift
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVARIANT="\"$name\"" "$@" attn_microbench.hip -o bin/mb_$name &
}
build wps2_acc
build wps2_fast -DSMML_FAST_MATH=1
build wps2_nodpp -DSMML_DPP_REDUCE=0
build wps1_acc -DSMML_FWD_WPS=1 -DSMML_BWD_WPS=1
build wps1_fast -DSMML_FWD_WPS=1 -DSMML_BWD_WPS=1 -DSMML_FAST_MATH=1
wait
ls -la bin
