#!/bin/bash
# developer tool: build a library variant with extra flags into tools/ablate/lib_<name>.so
# (set PROMP_BASE_FLAGS to replace the default -mllvm flags)
name=$1; shift
BASE=${PROMP_BASE_FLAGS--mllvm -amdgpu-mfma-vgpr-form}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize $BASE -shared -fPIC \
  promp_amd/csrc/promp_hip.hip -o tools/ablate/lib_$name.so -lrccl -Wno-pass-failed "$@"
