#!/bin/bash
# Collects the round's profile set on the GPU box into gpurun_out/ (copy what should be kept to profiles/rNN/):
#   kernel-trace stats of the default bench, eight single-counter PMC passes, the PCIe-inclusive and
#   ncnn-compat-tile bench lines.  Run as:  gpurun --timeout 1500 -- 'bash scripts/collect_profiles.sh'
# rocprofv3 gets the python interpreter itself after "--" (no env/bash hop), one counter per pass.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# which sources the profiled library was built from (profiles/traffic.json quotes it; bench.py compares: roofline.traffic_stale)
(cd $R && python3 -c "import json; from reve_amd import _lib; print(json.dumps(_lib.build_info()))") > $O/build_info.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt -o kt -- python3 $R/bench.py --steps 200 --min-timed-s 0 --no-pcie --no-cpu-baseline --no-options-leg --no-configs > $O/bench_kt.log 2>&1
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_ANY; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/prof_pmc_$c -o pmc -- python3 $R/bench.py --steps 30 --warmup 5 --min-timed-s 0 --no-pcie --no-cpu-baseline --no-options-leg --no-configs > $O/bench_pmc_$c.log 2>&1
done
# the layer-per-launch body kernel's traffic for comparison (the default run above fuses the body layers in pairs)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/prof_unfused_pmc_$c -o pmc -- python3 $R/bench.py --steps 30 --warmup 5 --min-timed-s 0 --no-pcie --no-cpu-baseline --no-options-leg --no-configs --fuse 0 --winograd 0 > $O/bench_unfused_pmc_$c.log 2>&1
done
# round 6: the default run above is the Winograd pair kernel (option "winograd" = auto); the DIRECT pair kernel (k_pair, REVE_WINOGRAD=0) in
# its own passes: traffic, MFMA busy share, LDS conflicts
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/prof_direct_pmc_$c -o pmc -- python3 $R/bench.py --steps 30 --warmup 5 --min-timed-s 0 --no-pcie --no-cpu-baseline --no-configs --no-options-leg --winograd 0 > $O/bench_direct_pmc_$c.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt_direct -o kt -- python3 $R/bench.py --steps 200 --min-timed-s 0 --no-pcie --no-cpu-baseline --no-options-leg --no-configs --winograd 0 > $O/bench_kt_direct.log 2>&1
# host stages (roctx ranges: upload / chain / download / wait) next to kernels and copies: one trace of the pinned-host ring
REVE_ROCTX=1 timeout 300 rocprofv3 --marker-trace --kernel-trace --memory-copy-trace --output-format csv -d $O/prof_marker -o mk -- python3 $R/scripts/ring_trace.py > $O/ring_trace.log 2>&1
cd $R
find $O/prof_kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
python3 scripts/pmc_summary.py $O/prof_pmc_* --kernel-stats $O/kernel_stats.csv --json $O/pmc_summary.json > $O/pmc_summary.txt 2>&1
python3 scripts/pmc_summary.py $O/prof_unfused_pmc_* --json $O/pmc_summary_unfused.json > $O/pmc_summary_unfused.txt 2>&1
find $O/prof_kt_direct -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_direct.csv \;
python3 scripts/pmc_summary.py $O/prof_direct_pmc_* --kernel-stats $O/kernel_stats_direct.csv --json $O/pmc_summary_direct.json > $O/pmc_summary_direct.txt 2>&1
python3 scripts/marker_summary.py $O/prof_marker > $O/marker_trace_summary.txt 2>&1
timeout 300 python3 bench.py --steps 300 --fuse 0 --winograd 0 --no-cpu-baseline --no-configs --min-timed-s 1 > $O/bench_unfused.json 2>/dev/null
python3 scripts/ab_pair.py > $O/ab_pair_1080p.txt 2>&1
W=3840 H=2160 N=10 python3 scripts/ab_pair.py > $O/ab_pair_4k.txt 2>&1
W=960 H=540 N=60 python3 scripts/ab_pair.py > $O/ab_pair_960x540.txt 2>&1
python3 scripts/ring_graph_ab.py > $O/ring_graph_ab.txt 2>&1
timeout 600 python3 bench.py --steps 1000 > $O/bench_full.json 2> $O/bench_full.err
timeout 300 python3 bench.py --steps 300 --tile 200 --no-cpu-baseline --min-timed-s 2 > $O/bench_tile200.json 2>/dev/null
for w in C3 C3-literal C5; do timeout 300 python3 bench.py --steps 200 --workload $w --no-cpu-baseline --min-timed-s 2 > $O/bench_$w.json 2>/dev/null; done
# the sizes of the reference's own assets and of BASELINE config 1: frames that share their kernel launches (option "batch"), and the same one per launch
for w in 960x540 640x480 256x256 100x100; do
  timeout 300 python3 bench.py --steps 200 --workload $w --no-cpu-baseline --min-timed-s 2 > $O/bench_$w.json 2>/dev/null
  timeout 300 python3 bench.py --steps 200 --workload $w --no-cpu-baseline --min-timed-s 2 --batch 0 > $O/bench_${w}_one_per_launch.json 2>/dev/null
done
python3 scripts/ab_batch.py > $O/ab_batch.txt 2>&1
python3 scripts/ab_wino.py > $O/ab_wino.txt 2>&1
for wh in "3840 2160" "960 540"; do set -- $wh; W=$1 H=$2 CHECK=0 N=20 ROUNDS=5 python3 scripts/ab_wino.py 2>&1 | grep -E "per-layer|pairs"; done > $O/ab_wino_sizes.txt
# same-box pairs of the bench line, direct (pinned) and the default (auto: Winograd)
for i in 1 2 3; do for w in 0 auto; do python3 bench.py --winograd $w --no-cpu-baseline --no-pcie --no-configs --no-options-leg --min-timed-s 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('winograd', '$w', d['value'], 'frames/s, roofline.frac', d['roofline']['frac'])"; done; done > $O/bench_winograd_pairs.txt
# issue costs of a lone wave (VALU / SALU / LDS beside MFMAs, MFMA 32x32x16, packed fp32, mix, register banks) and the sustained MFMA rate
# under the power cap (binaries built by hipcc in scripts/ubench/, see profiles/rNN/README.md)
[ -x scripts/ubench/valu_issue ] && timeout 300 scripts/ubench/valu_issue > $O/ubench_valu_issue.txt 2>&1
[ -x scripts/ubench/mfma_rate ] && timeout 300 scripts/ubench/mfma_rate > $O/ubench_mfma_rate.txt 2>&1
timeout 300 python3 bench.py --steps 300 --winograd 0 --no-cpu-baseline --min-timed-s 2 > $O/bench_direct.json 2>/dev/null
timeout 300 python3 bench.py --steps 125 --workload C4 --no-cpu-baseline --min-timed-s 2 > $O/bench_C4_1gpu.json 2>/dev/null
# the N > 1 line as the driver would launch it (no --workload: config 4's schedule), rehearsed on the one GPU over gloo: 2 and 8 ranks
REVE_BENCH_BACKEND=gloo timeout 600 python3 bench.py --gpus 2 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_N2_dryrun_1gpu_gloo.json 2>/dev/null
REVE_BENCH_BACKEND=gloo timeout 900 python3 bench.py --gpus 8 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_N8_dryrun_1gpu_gloo.json 2>/dev/null
# ablation table of the pair kernel with in-kernel clocks (variants built by scripts/ablate_pair.sh before the call, see profiles/rNN/README.md)
if ls reve_amd/ablp_p_*.so >/dev/null 2>&1; then
  (cd reve_amd && ROUNDS=5 python3 ../scripts/ab_pair_libs.py shipped=libreve_hip.so $(for f in ablp_p_*.so; do n=${f#ablp_p_}; echo ${n%.so}=$f; done)) > $O/ablation_table_pair.txt 2>&1
fi
if ls reve_amd/ablp_w_*.so >/dev/null 2>&1; then       # (KFILE=kernels_wino.hip variants)
  (cd reve_amd && WINO=1 ROUNDS=5 python3 ../scripts/ab_pair_libs.py shipped=libreve_hip.so $(for f in ablp_w_*.so; do n=${f#ablp_w_}; echo ${n%.so}=$f; done)) > $O/ablation_table_wino.txt 2>&1
fi
bash scripts/power_probe.sh > $O/power_probe.txt 2>&1
find $O/prof_kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
cat $O/kernel_stats.csv | cut -c1-150
cat $O/bench_full.json
