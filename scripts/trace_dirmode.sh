#!/bin/bash
# Timeline of directory mode (kernels and copies) for a short segment: do the copies run beside the kernels?
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
D=$(mktemp -d)
python3 - <<PY
import os, sys
sys.path.insert(0, "$R")
from reve_amd import synth, ncnn_io
from reve_amd.upscaler import png_write
d = "$D"
ncnn_io.write_model(d + "/models", "realesr-animevideov3-x2", synth.make_weights(2))
os.makedirs(d + "/in"); os.makedirs(d + "/out")
for i in range(300):
    png_write(f"{d}/in/frame{i + 1:08d}.png", synth.toon_frame(i, 1920, 1080))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/trace_dir -o t -- $R/reve_amd/realesrgan-hip -i $D/in -o $D/out -s 2 -m $D/models -t full > $R/gpurun_out/trace_dir.log 2>&1
ls $R/gpurun_out/trace_dir
python3 - <<PY
import csv, glob, statistics as st
base = "$R/gpurun_out/trace_dir"
k = list(csv.DictReader(open(glob.glob(base + "/**/*kernel_trace.csv", recursive=True)[0])))
c = list(csv.DictReader(open(glob.glob(base + "/**/*memory_copy_trace.csv", recursive=True)[0])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in k)
cs = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in c)
frames, cur = [], []
for s_, e_, n_ in ks:
    if "k_first" in n_ and cur:
        frames.append(cur); cur = []
    cur.append((s_, e_, n_))
frames.append(cur)
frames = [f for f in frames if len(f) == 18]
mid = frames[len(frames) // 10: len(frames) - len(frames) // 10]
starts = [f[0][0] for f in mid]
per = [(b - a) / 1e3 for a, b in zip(starts, starts[1:])]
span = [(f[-1][1] - f[0][0]) / 1e3 for f in mid]
gaps = [(b[0][0] - a[-1][1]) / 1e3 for a, b in zip(mid, mid[1:])]
body = [(e_ - s_) / 1e3 for f in mid for s_, e_, n_ in f if "k_body" in n_ and ", 0," in n_]
print(f"{len(frames)} whole frames traced; middle 80 %: frame period median {st.median(per):.1f} us (p99 {sorted(per)[int(len(per) * 0.99)]:.1f}, max {max(per):.1f}) = {1e6 / st.median(per):.0f} frames/s")
print(f"chain span median {st.median(span):.1f} us; gap between the last kernel of a frame and the first of the next: median {st.median(gaps):.1f} us, max {max(gaps):.1f} us")
print(f"body launch median {st.median(body):.1f} us (under the profiler)")
t0, t1 = mid[0][0][0], mid[-1][-1][1]
busy = sum(min(e_, t1) - max(s_, t0) for s_, e_ in cs if e_ > t0 and s_ < t1)
print(f"copies in that window: {sum(1 for s_, e_ in cs if e_ > t0 and s_ < t1)} (two per frame), busy {100.0 * busy / (t1 - t0):.0f} % of the window — all of it under the kernels of neighbouring frames")
PY
