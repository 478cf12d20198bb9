"""Copies the set collected by scripts/collect_profiles.sh from gpurun_out/ into profiles/<round>/ and rewrites
profiles/traffic.json (the PMC-derived HBM bytes per body launch that bench.py reports as roofline.traffic)."""
import json
import shutil
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
src, dst = "gpurun_out", f"profiles/{rnd}"
s = json.load(open(f"{src}/pmc_summary.json"))
import re
import os
t = {"algorithmic_bytes_per_launch": 530841600,
     "source": f"profiles/{rnd}/pmc_summary*.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; "
               "FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section)"}
if os.path.exists(f"{src}/build_info.json"):      # which kernel sources the profiled library was built from
    bi = json.load(open(f"{src}/build_info.json"))
    t["pair_src_sha256"], t["wino_src_sha256"] = bi.get("pair_src_sha256"), bi.get("wino_src_sha256")


def record(prefix, v):
    f, w = v["FETCH_SIZE"] * 1024 * 2, v["WRITE_SIZE"] * 1024
    t.update({f"{prefix}_hbm_bytes_per_launch": int(f + w), f"{prefix}_fetch_bytes_corrected_x2": int(f), f"{prefix}_write_bytes": int(w)})
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v:      # one v_mfma_f32_16x16x32_f16 keeps its pipe busy for 16 cycles
        t[f"{prefix}_mfma_instructions_per_launch_pmc"] = int(round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / 16))


# the default run (option "winograd" = auto) is the Winograd pair kernel; the direct pair kernel has its own passes (--winograd 0)
kw = [x for x in s if "k_wino" in x]
if kw:
    record("wino", s[kw[0]])
sd = json.load(open(f"{src}/pmc_summary_direct.json")) if os.path.exists(f"{src}/pmc_summary_direct.json") else s
kp = [x for x in sd if "k_pair" in x]
if kp:       # the fused pair: one activation read + one write per TWO layers (same algorithmic bytes per launch)
    record("pair", sd[kp[0]])
su = json.load(open(f"{src}/pmc_summary_unfused.json")) if os.path.exists(f"{src}/pmc_summary_unfused.json") else s
kb = [x for x in su if re.search(r"k_body<\d, 0,", x)]     # the body layers (k_body<ORDER, 2 / 3 / 4, ...> are conv_last)
if kb:
    f, w = su[kb[0]]["FETCH_SIZE"] * 1024 * 2, su[kb[0]]["WRITE_SIZE"] * 1024
    t.update({"body_hbm_bytes_per_launch": int(f + w), "fetch_bytes_corrected_x2": int(f), "write_bytes": int(w)})
json.dump(t, open("profiles/traffic.json", "w"), indent=1)
os.makedirs(dst, exist_ok=True)
for extra in ("pmc_summary_unfused.json", "pmc_summary_unfused.txt", "marker_trace_summary.txt", "bench_unfused.json", "ab_pair_1080p.txt",
              "ab_pair_4k.txt", "ab_pair_960x540.txt", "ring_graph_ab.txt", "ablation_table_pair.txt", "power_probe.txt",
              "bench_C4_1gpu.json", "bench_C4_2ranks_1gpu_gloo.json", "ab_batch.txt", "ab_wino.txt", "bench_winograd.json",
              "bench_960x540.json", "bench_640x480.json", "bench_256x256.json", "bench_100x100.json",
              "bench_960x540_one_per_launch.json", "bench_640x480_one_per_launch.json", "bench_256x256_one_per_launch.json",
              "bench_100x100_one_per_launch.json", "pmc_summary_winograd.txt", "pmc_summary_winograd.json", "ab_wino_sizes.txt",
              "bench_winograd_pairs.txt", "build_info.json", "pmc_summary_direct.json", "pmc_summary_direct.txt", "kernel_stats_direct.csv", "bench_direct.json",
              "bench_N2_dryrun_1gpu_gloo.json", "bench_N8_dryrun_1gpu_gloo.json", "kernel_stats_C2_direct.csv", "kernel_stats_C2_tile200.csv", "kernel_stats_C3.csv",
              "kernel_stats_C3_literal.csv", "kernel_stats_C5.csv", "ubench_valu_issue.txt", "ubench_mfma_rate.txt", "ablation_table_wino.txt"):
    # (gpurun_out/ is scratch that outlives rounds: only what THIS collection wrote — not older than its build_info.json — is installed)
    if os.path.exists(f"{src}/{extra}") and os.path.getmtime(f"{src}/{extra}") >= os.path.getmtime(f"{src}/build_info.json") - 60:
        shutil.copy(f"{src}/{extra}", f"{dst}/{extra}")
for a, b in [("kernel_stats.csv", "kernel_stats_bench_steps200.csv"), ("pmc_summary.json", "pmc_summary.json"),
             ("pmc_summary.txt", "pmc_summary.txt"), ("bench_full.json", "bench_steps1000_pcie.json"),
             ("bench_tile200.json", "bench_tile200.json"), ("bench_C3.json", "bench_C3_1080p_x4.json"),
             ("bench_C3-literal.json", "bench_C3literal_960x540_x4.json"), ("bench_C5.json", "bench_C5_4k_x2.json")]:
    shutil.copy(f"{src}/{a}", f"{dst}/{b}")
    if b.startswith("bench"):
        d = json.loads(open(f"{dst}/{b}").read().strip().splitlines()[-1])
        print(b, d["value"], d.get("pcie_inclusive_fps"), d["roofline"]["launch_us"], d["roofline"]["frac"],
              d.get("roofline_frac_whole_path"), (d.get("cpu_baseline") or {}).get("value"))
print(open("profiles/traffic.json").read())
for name, v in s.items():
    print(name[:48], {a: round(b) for a, b in v.items()})
