#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel per dispatch.

usage: pmc_summary.py gpurun_out/prof_pmc_*/  -> prints a table and (with --json PATH) writes it.
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
    if out_json:
        json.dump(res, open(out_json, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
