#!/bin/bash
# usage (on the GPU box, from repo root): scripts/pmc_ab.sh "<variants>" "<counter set 1>" "<counter set 2>" ...
# variant "default" = shipped library, otherwise reve_amd/abl_<variant>.so
R=$PWD; V="$1"; shift
cd /tmp && export TMPDIR=/tmp
i=0
for C in "$@"; do
  i=$((i+1))
  for v in $V; do
    if [ "$v" = default ]; then unset REVE_HIP_LIB; else export REVE_HIP_LIB=$R/reve_amd/abl_$v.so; fi
    N=6 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmcab/$v/set$i -- python3 $R/scripts/time_body.py > $R/gpurun_out/pmcab_${v}_$i.log 2>&1
  done
done
cd $R
for v in $V; do echo "=== $v"; python3 scripts/pmc_summary.py gpurun_out/pmcab/$v/set* | awk '/k_body/{f=1;next} /^[a-z]/{f=0} f'; done
