#!/bin/bash
# rocprofv3 kernel-trace stats of every BASELINE configuration bench.py times (the `configs` legs of its one line) and of the direct
# kernels pinned (--winograd 0; the default evaluation is auto = Winograd), one short profiled run each: gpurun_out/kernel_stats_<config>.csv (copy to profiles/rNN/).
#   gpurun --timeout 1200 -- 'bash scripts/collect_config_traces.sh'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {   # name, bench arguments...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg_$name -o kt -- python3 $R/bench.py --steps 60 --min-timed-s 0 --no-pcie --no-cpu-baseline --no-options-leg --no-configs "$@" > $O/bench_cfg_$name.log 2>&1
  find $O/prof_cfg_$name -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$name.csv \;
  echo "== $name"; cut -d, -f1-5 $O/kernel_stats_$name.csv | head -6
}
run C2_tile200 --tile 200
run C3 --workload C3
run C3_literal --workload C3-literal
run C5 --workload C5 --steps 20
run C2_direct --winograd 0
