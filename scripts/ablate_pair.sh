#!/bin/bash
# Variants of the fused-pair kernel (kernels_pair.hip) as separate libraries reve_amd/ablp_<name>.so (STAMPS / ABLP_* are
# timing-only).  Usage: scripts/ablate_pair.sh NAME "-D..." [NAME2 "-D..."]...; compare them with scripts/ab_pair_libs.py.
set -e
cd "$(dirname "$0")/../reve_amd/csrc"
mkdir -p build
make -s all >/dev/null
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -DREVE_DIAGNOSTIC_BUILD $flags -c kernels_pair.hip -o build/kernels_pair_$name.o
  objs="build/kernels.hip.o build/kernels_first.hip.o build/engine.cpp.o build/model.cpp.o build/capi.cpp.o build/png.cpp.o build/fastdeflate.cpp.o build/dirmode.cpp.o build/hostbind.cpp.o"
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../ablp_$name.so build/kernels_pair_$name.o $objs -lz -ldl
  echo built ablp_$name.so
done
