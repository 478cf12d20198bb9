"""Times the kernels of the frame chain at 1080p x2 for one library build (REVE_HIP_LIB) and prints
per-kernel averages from the library's own HIP-event brackets.  Several builds are compared by running
this script once per build, interleaved rounds: scripts/ab.sh."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reve_amd import synth, ncnn_io
from reve_amd.upscaler import Upscaler
scale = int(os.environ.get("SCALE", "2")); W = int(os.environ.get("W", "1920")); H = int(os.environ.get("H", "1080"))
n = int(os.environ.get("N", "60"))
w = synth.make_weights(scale)
up = Upscaler(scale, param=ncnn_io.build_param_text(scale).encode(), bin=ncnn_io.build_bin(w))
src = torch.from_numpy(synth.noise_frame(0, W, H)).cuda()
dst = torch.empty((H * scale, W * scale, 3), dtype=torch.uint8, device="cuda")
for _ in range(5):
    up.upscale_device(src.data_ptr(), W, H, dst.data_ptr())
up.sync(); up.set_profiling(True); up.reset_stats()
res = []
for rnd in range(3):
    up.reset_stats()
    t0 = time.perf_counter()
    for _ in range(n):
        up.upscale_device(src.data_ptr(), W, H, dst.data_ptr())
    up.sync(); dt = time.perf_counter() - t0
    st = up.stats()
    res.append((dt / n * 1e3, st["body_ms_total"] / max(st["body_launches"], 1) * 1e3))
print(os.environ.get("REVE_HIP_LIB", "default").split("/")[-1], " ".join(f"[{a:.3f} ms/frame, body {b:.1f} us]" for a, b in res), flush=True)
