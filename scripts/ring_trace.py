"""A short run of the pinned-host ring (reve_submit / reve_wait, depth 3) for a rocprofv3 timeline: with REVE_ROCTX=1 the
library's roctx ranges (reve:upload / reve:chain / reve:download / reve:wait) appear in the marker trace next to the kernels
and the H2D / D2H copies they overlap (scripts/collect_profiles.sh, scripts/marker_summary.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reve_amd import synth, ncnn_io
from reve_amd.upscaler import Upscaler, pinned_array
S, W, H = 2, 1920, 1080
w = synth.make_weights(S)
up = Upscaler(S, param=ncnn_io.build_param_text(S).encode(), bin=ncnn_io.build_bin(w))
hin = [pinned_array((H, W, 3)) for _ in range(3)]
hout = [pinned_array((H * S, W * S, 3)) for _ in range(3)]
for k in range(3):
    hin[k][...] = synth.noise_frame(k, W, H)
n = int(os.environ.get("N", "120"))
for i in range(n):
    if i >= 3:
        up.wait()
    up.submit(i, hin[i % 3], hout[i % 3])
for _ in range(3):
    up.wait()
print("ring trace run done:", n, "frames")
