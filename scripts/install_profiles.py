"""Copies the set collected by scripts/collect_profiles.sh from gpurun_out/ into profiles/<round>/ and rewrites
profiles/traffic.json (the PMC-derived HBM bytes per body launch that bench.py reports as roofline.traffic)."""
import json
import shutil
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
src, dst = "gpurun_out", f"profiles/{rnd}"
s = json.load(open(f"{src}/pmc_summary.json"))
import re
k = [x for x in s if re.search(r"k_body<\d, 0,", x)][0]     # the body layers (k_body<ORDER, 2 / 3 / 4, ...> are conv_last)
f, w = s[k]["FETCH_SIZE"] * 1024 * 2, s[k]["WRITE_SIZE"] * 1024
json.dump({"body_hbm_bytes_per_launch": int(f + w), "fetch_bytes_corrected_x2": int(f), "write_bytes": int(w),
           "algorithmic_bytes_per_launch": 530841600,
           "source": f"profiles/{rnd}/pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; "
                     "FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section)"}, open("profiles/traffic.json", "w"), indent=1)
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
