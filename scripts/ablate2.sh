#!/bin/bash
# Variants of the shipped body kernel (kernels_body2.hip) as separate libraries reve_amd/abl2_<name>.so (timing only for the
# ABL2_* switches: outputs are wrong).  Usage: scripts/ablate2.sh NAME "-D..." [NAME2 "-D..."]...; compare them in one
# process with scripts/ab_libs.py.
set -e
cd "$(dirname "$0")/../reve_amd/csrc"
mkdir -p build
make -s all >/dev/null
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 $flags -c kernels_body2.hip -o build/kernels_body2_$name.o
  objs=$(ls build/*.hip.o build/*.cpp.o | grep -v kernels_body2.hip.o)
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl2_$name.so build/kernels_body2_$name.o $objs -lz
  echo built abl2_$name.so
done
