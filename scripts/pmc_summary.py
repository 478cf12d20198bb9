#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel per dispatch.

usage: pmc_summary.py gpurun_out/prof_pmc_*/ [--kernel-stats kernel_stats.csv] [--json PATH]  -> prints a table and writes it.
Besides the raw counters it prints the two derived figures BASELINE's north_star asks for, per kernel:
  achieved HBM GB/s = (FETCH_SIZE x 2 + WRITE_SIZE) bytes / the kernel's average duration — from the --kernel-stats csv of a
                      plain kernel-trace run if given (counter passes run slower: profiled clocks), else from the counter passes;
                      against 8,000 GB/s
  MFMA utilisation  = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8 XCDs): the share of the launch in which
                      a matrix pipe is busy, summed over the chip; and the executed MFMA FLOP/s (busy cycles x 1,024 FLOP per cycle
                      and SIMD for the 16-bit formats / duration) against the 2.5 PFLOP/s dense peak.
FETCH_SIZE/WRITE_SIZE are in KiB-ish units of 1 KB... rocprofv3 reports them in KB (1024 B); on gfx950
FETCH_SIZE under-reports wide streaming reads by 2x (MI355X_MICROARCH.md §HBM): both raw and
corrected values are printed.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    out_json = None
    if "--json" in sys.argv:
        out_json = sys.argv[sys.argv.index("--json") + 1]
        args = [a for a in args if a != out_json]
    kstats = {}
    if "--kernel-stats" in sys.argv:
        kpath = sys.argv[sys.argv.index("--kernel-stats") + 1]
        args = [a for a in args if a != kpath]
        for r in csv.DictReader(open(kpath)):
            kstats[r["Name"]] = float(r["AverageNs"])
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for d in args:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    res = {}
    for k in sorted(acc):
        if "rocclr" in k:
            continue
        res[k] = {c: sum(v) / len(v) for c, v in acc[k].items()}
        res[k]["_dispatches"] = max(len(v) for v in acc[k].values())
        res[k]["_mean_ns_profiled"] = sum(dur[k]) / len(dur[k])
    for k, v in res.items():
        print(k)
        for c, x in sorted(v.items()):
            print(f"    {c:32s} {x:18.2f}")
        if "FETCH_SIZE" in v:
            print(f"    {'FETCH bytes (x1024 x2 corr.)':32s} {v['FETCH_SIZE'] * 1024 * 2:18.0f}")
        if "WRITE_SIZE" in v:
            print(f"    {'WRITE bytes (x1024)':32s} {v['WRITE_SIZE'] * 1024:18.0f}")
        ns = kstats.get(k, v["_mean_ns_profiled"])
        src = "kernel-trace run" if k in kstats else "counter passes"
        v["_duration_ns_used"] = ns
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            v["_hbm_gbps"] = (v["FETCH_SIZE"] * 2048 + v["WRITE_SIZE"] * 1024) / ns
            v["_hbm_frac_of_8000"] = v["_hbm_gbps"] / 8000.0
            print(f"    {'=> achieved HBM GB/s':32s} {v['_hbm_gbps']:18.1f}   = {v['_hbm_frac_of_8000']:.3f} of 8,000 GB/s ({ns / 1e3:.1f} us per launch, {src})")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
            v["_mfma_util"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * 256 * v["GRBM_GUI_ACTIVE"] / 8)
            print(f"    {'=> MFMA utilisation':32s} {v['_mfma_util']:18.3f}   (busy cycles / (1,024 SIMDs x GRBM_GUI_ACTIVE / 8))")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v:
            v["_mfma_tflops_executed"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] * 1024 / ns / 1e3
            v["_mfma_frac_of_2500"] = v["_mfma_tflops_executed"] / 2500.0
            print(f"    {'=> executed MFMA TFLOP/s':32s} {v['_mfma_tflops_executed']:18.1f}   = {v['_mfma_frac_of_2500']:.3f} of 2,500 TFLOP/s dense fp16 ({src})")
    if out_json:
        json.dump(res, open(out_json, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
