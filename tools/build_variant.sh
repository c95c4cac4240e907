#!/bin/bash
# developer tool: build a library variant with extra flags into tools/ablate/lib_<name>.so
# (set PROMP_BASE_FLAGS to add -mllvm flags)
name=$1; shift
BASE=${PROMP_BASE_FLAGS-}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize $BASE -shared -fPIC \
  promp_amd/csrc/promp_hip.hip -o tools/ablate/lib_$name.so -lrccl -Wno-pass-failed "$@"
