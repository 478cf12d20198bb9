#!/bin/bash
# Variants of the fused-pair kernel (kernels_pair.hip; KFILE=kernels_last.hip: of the conv_last strip kernel; KFILE=kernels_wino.hip:
# of the Winograd pair kernel, compare with WINO=1 scripts/ab_pair_libs.py) as separate
# libraries reve_amd/ablp_<name>.so (STAMPS / ABLP_* / KL_ABL_* are timing-only).
# Usage: [KFILE=...] scripts/ablate_pair.sh NAME "-D..." [NAME2 "-D..."]...; compare them with scripts/ab_pair_libs.py.
set -e
cd "$(dirname "$0")/../reve_amd/csrc"
mkdir -p build
make -s all >/dev/null
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  kfiles=${KFILE:-kernels_pair.hip}          # (one file or several, all compiled with the variant's flags)
  vobjs=""
  for kf in $kfiles; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 $([ $kf = kernels_wino.hip ] && echo -fno-slp-vectorize) -DREVE_DIAGNOSTIC_BUILD $flags -c $kf -o build/variant_${name}_$kf.o
    vobjs="$vobjs build/variant_${name}_$kf.o"
  done
  objs=""
  # (every source of the library — the Makefile's SRCS — except the variant's own files)
  for f in $(sed -n 's/^SRCS := //p' Makefile); do
    case " $kfiles " in *" $f "*) ;; *) objs="$objs build/$f.o";; esac
  done
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../ablp_$name.so $vobjs $objs -lz -ldl
  echo built ablp_$name.so
done
