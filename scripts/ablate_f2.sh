#!/bin/bash
# like ablate.sh but varies kernels_f2.hip (fused path); use with REVE_FUSED=1
set -e
cd "$(dirname "$0")/../reve_amd/csrc"
mkdir -p build
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c kernels_f2.hip -o build/kernels_f2_$name.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl_$name.so build/kernels.hip.o build/kernels_first.hip.o build/kernels_last.hip.o build/kernels_exp.hip.o build/kernels_f2_$name.o build/engine.cpp.o build/model.cpp.o build/capi.cpp.o build/png.cpp.o build/dirmode.cpp.o -lz
  echo built abl_$name.so
done
