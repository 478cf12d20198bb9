"""Fused-pair body kernel against the layer-per-launch path: same library, two contexts, interleaved rounds in ONE process on
one device (rule 24).  Prints per-layer device time of the body chain (HIP events) and whole-frame rate for each.
env: N (frames per round, 30), ROUNDS (7), W, H."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from reve_amd import synth, ncnn_io
from reve_amd.upscaler import Upscaler
S = 2
W, H = int(os.environ.get("W", "1920")), int(os.environ.get("H", "1080"))
n = int(os.environ.get("N", "30")); rounds = int(os.environ.get("ROUNDS", "9"))
w = synth.make_weights(S)
p, b = ncnn_io.build_param_text(S).encode(), ncnn_io.build_bin(w)
src = torch.from_numpy(synth.noise_frame(0, W, H)).cuda()
dst = torch.empty((H * S, W * S, 3), dtype=torch.uint8, device="cuda")
ups = {}
for name, fused in (("layer-per-launch", 0), ("fused-pairs", 1)):
    up = Upscaler(S, param=p, bin=b)
    up.set_option("winograd", 0)          # this A/B is about the DIRECT pair kernel (the library's default evaluation is auto)
    up.set_option("fuse_pairs", fused)
    for _ in range(3):
        up.upscale_device(src.data_ptr(), W, H, dst.data_ptr())
    up.sync()
    up.set_profiling(True)
    ups[name] = up
names = list(ups)
body = {k: [] for k in names}; fps = {k: [] for k in names}; last = {}
for r in range(rounds):
    for k in (names if r % 2 == 0 else names[::-1]):
        up = ups[k]
        up.reset_stats()
        t0 = time.perf_counter()
        for _ in range(n):
            up.upscale_device(src.data_ptr(), W, H, dst.data_ptr())
        up.sync()
        dt = time.perf_counter() - t0
        st = up.stats()
        body[k].append(st["body_ms_total"] / max(st["body_launches"], 1) * 1e3)
        last.setdefault(k, []).append(st["last_ms_total"] / max(st["frames_timed"], 1) * 1e3)
        fps[k].append(n / dt)
for k in names:
    v = sorted(body[k]); f = sorted(fps[k])
    lv = sorted(last[k])
    print(f"{k:18s} body per layer median {v[len(v) // 2]:7.2f} us (min {v[0]:7.2f}, max {v[-1]:7.2f}); conv_last {lv[len(lv) // 2]:6.2f} us; frames/s median {f[len(f) // 2]:7.1f} (max {f[-1]:7.1f})", flush=True)
a = sorted(body[names[0]])
for k in names[1:]:
    c = sorted(body[k])
    print(f"{k} / {names[0]} per-layer time: {c[len(c) // 2] / a[len(a) // 2]:.4f}")
img = synth.noise_frame(3, W, H)
x, y = ups["fused-pairs"].upscale(img), ups["layer-per-launch"].upscale(img)
print("fused-pairs vs layer-per-launch output bytes identical:", bool(np.array_equal(x, y)))
