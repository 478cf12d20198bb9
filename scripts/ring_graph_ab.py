"""The submit/wait ring with and without the captured hipGraph (option "graph"), fused pairs on: wall-clock frames/s of the
pinned-host pipeline and host CPU time spent in reve_submit per frame, interleaved rounds in one process.
env: N (frames per round, 300), ROUNDS (5)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reve_amd import synth, ncnn_io
from reve_amd.upscaler import Upscaler, pinned_array
S, W, H = 2, 1920, 1080
n = int(os.environ.get("N", "300")); rounds = int(os.environ.get("ROUNDS", "5"))
w = synth.make_weights(S)
p, b = ncnn_io.build_param_text(S).encode(), ncnn_io.build_bin(w)
hin = [pinned_array((H, W, 3)) for _ in range(3)]
hout = [pinned_array((H * S, W * S, 3)) for _ in range(3)]
for k in range(3):
    hin[k][...] = synth.noise_frame(k, W, H)
ups = {}
for name, g in (("launches", 0), ("graph", 1)):
    up = Upscaler(S, param=p, bin=b)
    up.set_option("fuse_pairs", 1)
    up.set_option("graph", g)
    ups[name] = up
def run(up, n):
    t_sub = 0.0
    t0 = time.perf_counter()
    for i in range(n):
        if i >= 3:
            up.wait()
        a = time.perf_counter()
        up.submit(i, hin[i % 3], hout[i % 3])
        t_sub += time.perf_counter() - a
    for _ in range(3):
        up.wait()
    return n / (time.perf_counter() - t0), t_sub / n * 1e6
for up in ups.values():
    run(up, 20)
res = {k: [] for k in ups}
names = list(ups)
for r in range(rounds):
    for k in (names if r % 2 == 0 else names[::-1]):
        res[k].append(run(ups[k], n))
for k in names:
    f = sorted(x[0] for x in res[k]); s = sorted(x[1] for x in res[k])
    print(f"{k:9s} ring {f[len(f) // 2]:7.1f} frames/s (max {f[-1]:7.1f}); reve_submit takes {s[len(s) // 2]:6.1f} us of host time per frame", flush=True)
