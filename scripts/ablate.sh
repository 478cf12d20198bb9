#!/bin/bash
# Builds ablation variants of the kernels (timing-only: outputs are wrong) into gpurun_out-independent
# files reve_amd/abl_<name>.so.  Usage: scripts/ablate.sh NAME "-DABL_..." [NAME2 "-D..."]...
set -e
cd "$(dirname "$0")/../reve_amd/csrc"
mkdir -p build
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c kernels.hip -o build/kernels_$name.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl_$name.so build/kernels_$name.o build/kernels_f2.hip.o build/kernels_exp.hip.o build/engine.cpp.o build/model.cpp.o build/capi.cpp.o build/png.cpp.o build/dirmode.cpp.o -lz
  echo built abl_$name.so
done
