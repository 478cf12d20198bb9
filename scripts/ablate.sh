#!/bin/bash
# Variants of the shipped body kernel (kernels.hip) as separate libraries reve_amd/abl_<name>.so (timing only for the
# ABL2_* switches: outputs are wrong).  Usage: scripts/ablate.sh NAME "-D..." [NAME2 "-D..."]...; compare them in one
# process with scripts/ab_libs.py.
set -e
cd "$(dirname "$0")/../reve_amd/csrc"
mkdir -p build
make -s all >/dev/null
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -DREVE_DIAGNOSTIC_BUILD $flags -c kernels.hip -o build/kernels_$name.o
  objs="build/kernels_first.hip.o build/engine.cpp.o build/model.cpp.o build/capi.cpp.o build/png.cpp.o build/fastdeflate.cpp.o build/dirmode.cpp.o build/hostbind.cpp.o"
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl_$name.so build/kernels_$name.o $objs -lz -ldl
  echo built abl_$name.so
done
