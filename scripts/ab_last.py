"""conv_last x2 as the rolling-strip kernel (kernels_last.hip, option "strip_last") against the tile kernel: same library, two
contexts, interleaved rounds in ONE process on one device.  First checks that the output bytes are identical on a set of
shapes, then prints conv_last's device time per frame (HIP events around the launch) and the whole-frame rate for each.
env: N (frames per round, 30), ROUNDS (9), W, H."""
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
ups = {}
for name, strip in (("tile kernel", 0), ("strip kernel", 1)):
    up = Upscaler(S, param=p, bin=b)
    up.set_option("strip_last", strip)
    ups[name] = up
names = list(ups)
bad = 0
for (sw, sh) in [(96, 64), (60, 33), (61, 17), (1, 1), (3, 2), (121, 35), (180, 7), (200, 131), (640, 360), (62, 9), (63, 40), (124, 18),
                 (125, 21), (1240, 200), (16100, 20), (1920, 1080), (33, 1000)]:
    img = synth.noise_frame(sw * 1000 + sh, sw, sh)
    x, y = ups[names[0]].upscale(img), ups[names[1]].upscale(img)
    if not np.array_equal(x, y):
        bad += 1
        d = np.argwhere(x != y)
        print(f"MISMATCH {sw}x{sh}: {len(d)} bytes differ, first {d[:6].tolist()}, rows {sorted(set(d[:, 0].tolist()))[:12]}, cols {sorted(set(d[:, 1].tolist()))[:12]}", flush=True)
print("identical output bytes on all shapes" if not bad else f"{bad} shapes differ", flush=True)
src = torch.from_numpy(synth.noise_frame(0, W, H)).cuda()
dst = torch.empty((H * S, W * S, 3), dtype=torch.uint8, device="cuda")
for up in ups.values():
    for _ in range(3):
        up.upscale_device(src.data_ptr(), W, H, dst.data_ptr())
    up.sync()
    up.set_profiling(True)
last = {k: [] for k in names}; fps = {k: [] for k in names}
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
        last[k].append(st["last_ms_total"] / max(st["frames_timed"], 1) * 1e3)
        fps[k].append(n / dt)
for k in names:
    v = sorted(last[k]); f = sorted(fps[k])
    print(f"{k:14s} conv_last median {v[len(v) // 2]:7.2f} us (min {v[0]:7.2f}, max {v[-1]:7.2f}); frames/s median {f[len(f) // 2]:7.1f} (max {f[-1]:7.1f})", flush=True)
