#!/bin/bash
# PMC passes of the drop-in default (the binary's tiling: 200-pixel tiles + 10-pixel apron, what an unmodified reve gets): the canvas
# instantiations k_wino<true, true>, k_last_strip<2, true> and conv_first over planes.  -> gpurun_out/pmc_summary_tile200.{txt,json}
#   gpurun --timeout 900 -- 'bash scripts/collect_tile200_pmc.sh'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_t200_kt -o kt -- python3 $R/bench.py --tile 200 --steps 100 --min-timed-s 0 --no-pcie --no-cpu-baseline --no-options-leg --no-configs > $O/bench_t200_kt.log 2>&1
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/prof_t200_pmc_$c -o pmc -- python3 $R/bench.py --tile 200 --steps 30 --warmup 5 --min-timed-s 0 --no-pcie --no-cpu-baseline --no-options-leg --no-configs > $O/bench_t200_pmc_$c.log 2>&1
done
cd $R
find $O/prof_t200_kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_tile200.csv \;
python3 scripts/pmc_summary.py $O/prof_t200_pmc_* --kernel-stats $O/kernel_stats_tile200.csv --json $O/pmc_summary_tile200.json > $O/pmc_summary_tile200.txt 2>&1
cat $O/pmc_summary_tile200.txt
