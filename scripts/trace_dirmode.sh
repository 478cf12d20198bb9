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
for i in range(80):
    png_write(f"{d}/in/frame{i + 1:08d}.png", synth.toon_frame(i, 1920, 1080))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/trace_dir -o t -- $R/reve_amd/realesrgan-hip -i $D/in -o $D/out -s 2 -m $D/models -t full > $R/gpurun_out/trace_dir.log 2>&1
ls $R/gpurun_out/trace_dir
python3 - <<PY
import csv, glob
base = "$R/gpurun_out/trace_dir"
k = list(csv.DictReader(open(glob.glob(base + "/**/*kernel_trace.csv", recursive=True)[0])))
c = list(csv.DictReader(open(glob.glob(base + "/**/*memory_copy_trace.csv", recursive=True)[0])))
print(len(k), "kernels", len(c), "copies; copy columns:", list(c[0].keys()))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:30]) for r in k)
cs = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", r.get("Name", "?"))) for r in c)
t0 = ks[len(ks) // 2][0]
print("kernels around the middle:")
for s, e, n in ks[len(ks) // 2: len(ks) // 2 + 40]:
    print(f"  {(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f} us  {n}")
print("copies in that window:")
for s, e, n in cs:
    if t0 <= s <= ks[len(ks) // 2 + 40][1]:
        print(f"  {(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f} us  {n} ({(e - s) / 1e3:.0f} us)")
PY
