#!/usr/bin/env python3
"""Instruction mix of the MFMA-carrying loops of a kernel, from `hipcc -S` output: what a lone wave pays for a step.

A 512-register wave is alone on its SIMD and issues ONE instruction of any class per 4 cycles (an MFMA 16x16x32 takes two such
turns; profiles/r04/ubench_valu_issue.txt), so below the MFMA pipe's own time (16 cycles each) a step costs
8 x MFMAs + 4 x everything else — the count printed here.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 [-fno-slp-vectorize] --cuda-device-only -S K.hip -o K.s
    python scripts/isa_loop_mix.py K.s [kernel-name-substring]
"""
import collections
import re
import sys


def main():
    lines = open(sys.argv[1]).read().split("\n")
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and want in l]
    for st in starts:
        end = next(i for i in range(st, len(lines)) if "s_endpgm" in lines[i])
        labels = {}
        for i in range(st, end):
            m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
            if m:
                labels[m.group(1)] = i
        loops = []
        for i in range(st, end):
            m = re.match(r"\s+s_cbranch_\w+ (\.LBB\d+_\d+)", lines[i]) or re.match(r"\s+s_branch (\.LBB\d+_\d+)", lines[i])
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i))
        print(lines[st].split(":")[0])
        # innermost MFMA-carrying loops only
        loops = [(a, b) for a, b in loops if any("v_mfma" in l for l in lines[a:b])]
        inner = [(a, b) for a, b in loops if not any((c, d) != (a, b) and a <= c and d <= b for c, d in loops)]
        for a, b in inner:
            c = collections.Counter()
            for l in lines[a:b + 1]:
                m = re.match(r"\s+([a-z_0-9]+)", l)
                if m and not l.strip().startswith((";", ".")):
                    c[m.group(1)] += 1
            mf = sum(v for k, v in c.items() if k.startswith("v_mfma"))
            tot = sum(c.values())
            cls = collections.Counter()
            for k, v in c.items():
                cls["mfma" if k.startswith("v_mfma") else "valu" if k.startswith("v_") else "salu" if k.startswith("s_") else
                    "lds" if k.startswith("ds_") else "vmem"] += v
            print(f"  loop at +{a - st}..+{b - st}: {tot} instructions, {dict(cls)}; issue model 8 x {mf} + 4 x {tot - mf} = {8 * mf + 4 * (tot - mf)} cycles"
                  f" (MFMA pipe {16 * mf})")
            print("    " + ", ".join(f"{k} {v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])[:32]))


if __name__ == "__main__":
    main()
