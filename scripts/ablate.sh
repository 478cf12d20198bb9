#!/bin/bash
# Builds ablation / parameter variants of the kernels (possibly timing-only: outputs may be wrong) into
# reve_amd/abl_<name>.so; select one with REVE_HIP_LIB.  Usage: scripts/ablate.sh NAME "-D..." [NAME2 "-D..."]...
set -e
cd "$(dirname "$0")/../reve_amd/csrc"
mkdir -p build
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 $flags -c kernels.hip -o build/kernels_$name.o
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 $flags -c kernels_body2.hip -o build/kernels_body2_$name.o
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c kernels_last.hip -o build/kernels_last_$name.o
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 $flags -c kernels_first.hip -o build/kernels_first_$name.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl_$name.so build/kernels_$name.o build/kernels_body2_$name.o build/kernels_first_$name.o build/kernels_last_$name.o build/kernels_f2.hip.o build/kernels_exp.hip.o build/engine.cpp.o build/model.cpp.o build/capi.cpp.o build/png.cpp.o build/dirmode.cpp.o -lz
  echo built abl_$name.so
done
