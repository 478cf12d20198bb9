"""First-contact GPU script: layer-by-layer parity vs the oracle + a quick timing."""
import sys, os, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reve_amd import synth, ncnn_io
from reve_amd.upscaler import Upscaler
from oracle import ref

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 2
w = synth.make_weights(scale)
param = ncnn_io.build_param_text(scale).encode()
binb = ncnn_io.build_bin(w)
up = Upscaler(scale, param=param, bin=binb)
img = synth.toon_frame(1, 70, 45)
for layer in (0, 1, 2, 16):
    g = up.debug_layer(img, layer)
    o = ref.layer(w, img, layer)
    d = np.abs(g - o)
    print(f"layer {layer}: max abs diff {d.max():.6f}  frac>0 {(d>0).mean():.5f}  ref absmax {np.abs(o).max():.3f}", flush=True)
out = up.upscale(img)
exp = ref.upscale(w, img)
d = np.abs(out.astype(int) - exp.astype(int))
print("full x%d: max LSB %d, n_diff %d / %d" % (scale, d.max(), (d > 0).sum(), d.size), flush=True)
# timing at 1080p, device-resident
import torch
W, H = 1920, 1080
src = torch.from_numpy(synth.noise_frame(0, W, H)).cuda()
dst = torch.empty((H * scale, W * scale, 3), dtype=torch.uint8, device="cuda")
up.set_profiling(True)
for _ in range(3):
    up.upscale_device(src.data_ptr(), W, H, dst.data_ptr())
up.sync(); up.reset_stats()
t0 = time.time(); n = 20
for _ in range(n):
    up.upscale_device(src.data_ptr(), W, H, dst.data_ptr())
up.sync(); dt = time.time() - t0
st = up.stats()
print(f"1080p x{scale}: {n/dt:.1f} fps, {dt/n*1e3:.3f} ms/frame; body avg {st['body_ms_total']/max(st['body_launches'],1)*1e3:.1f} us; stats {st}", flush=True)
