#!/bin/bash
# HIP API time of a cold 3-frame directory call through the executable: where do the first-call milliseconds go?
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
f = synth.toon_frame(0, 1920, 1080)
for i in range(3):
    png_write(f"{d}/in/frame{i + 1:08d}.png", f)
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --stats --output-format csv -d $R/gpurun_out/trace_startup -o t -- $R/reve_amd/realesrgan-hip -i $D/in -o $D/out -s 2 -m $D/models -t full > $R/gpurun_out/trace_startup.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/trace_startup/**/*hip_api_stats.csv", recursive=True)
print(f)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:25]:
    print({k: r[k] for k in list(r)[:6]})
PY
