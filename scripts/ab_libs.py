"""Times the body launch of several library builds in ONE process on one device, interleaved rounds (rule 24), and, for
-DSTAMPS builds, reads the in-kernel clock and cycles per tile.  Usage: python scripts/ab_libs.py name=path.so ...
env: N (frames per round, 30), ROUNDS (5), TILE."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from reve_amd import synth, ncnn_io, _lib
from reve_amd.upscaler import Upscaler
S, W, H = 2, 1920, 1080
n = int(os.environ.get("N", "30")); rounds = int(os.environ.get("ROUNDS", "5")); tile = int(os.environ.get("TILE", "0"))
w = synth.make_weights(S)
p, b = ncnn_io.build_param_text(S).encode(), ncnn_io.build_bin(w)
src = torch.from_numpy(synth.noise_frame(0, W, H)).cuda()
dst = torch.empty((H * S, W * S, 3), dtype=torch.uint8, device="cuda")
ups, libs = {}, {}
for arg in sys.argv[1:]:
    name, path = arg.split("=", 1)
    _lib._lib = None
    _lib.LIB_PATH = os.path.abspath(path)
    ups[name] = Upscaler(S, param=p, bin=b, tile=tile)
    libs[name] = _lib._lib
    for _ in range(3):
        ups[name].upscale_device(src.data_ptr(), W, H, dst.data_ptr())
    ups[name].sync()
    ups[name].set_profiling(True)
names = list(ups)
res = {k: [] for k in names}
for r in range(rounds):
    for k in (names if r % 2 == 0 else names[::-1]):
        up = ups[k]
        up.reset_stats()
        for _ in range(n):
            up.upscale_device(src.data_ptr(), W, H, dst.data_ptr())
        up.sync()
        st = up.stats()
        res[k].append(st["body_ms_total"] / max(st["body_launches"], 1) * 1e3)
tiles = 4080 / 256.0
for k in names:
    v = sorted(res[k])
    line = f"{k:14s} body launch median {v[len(v) // 2]:7.2f} us (min {v[0]:7.2f}, max {v[-1]:7.2f})"
    lib = libs[k]
    if hasattr(lib, "reve_debug_read_stamps2"):
        for _ in range(n):      # the stamps that count are those of the last launch after a stretch of this variant alone
            ups[k].upscale_device(src.data_ptr(), W, H, dst.data_ptr())
        ups[k].sync()
        buf = (C.c_ulonglong * (1024 * 8))()
        lib.reve_debug_read_stamps2(buf, 1024 * 8)
        a = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.float64)
        a = a[a[:, 1] > 0]
        wall = (a[:, 3] - a[:, 2]) / 100.0
        clk = (a[:, 5] - a[:, 4]) / wall
        line += (f" | in-kernel {wall.mean():6.1f} us (slowest wave {wall.max():6.1f}), clock {np.median(clk):5.0f} MHz ({clk.min():.0f}-{clk.max():.0f}), "
                 f"{a[:, 1].mean() / tiles:6.0f} cycles/tile, barrier wait {100 * a[:, 0].sum() / a[:, 1].sum():4.1f} %")
    print(line, flush=True)
