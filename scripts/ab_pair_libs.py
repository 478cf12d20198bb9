"""Times the fused-pair body chain of several library builds (scripts/ablate_pair.sh) and the layer-per-launch chain of the
first one in ONE process on one device, interleaved rounds (rule 24); for -DSTAMPS builds it reads the in-kernel clock, the
cycles a wave spends per launch and its share waiting at the step barriers.
Usage: python scripts/ab_pair_libs.py name=path.so ...   env: N (frames per round, 30), ROUNDS (5), FUSE (1; 0: every library
runs one layer per launch), TILE (0), WINO (0; 1: the fused pairs are the Winograd kernel's, kernels_wino.hip)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from reve_amd import synth, ncnn_io, _lib
from reve_amd.upscaler import Upscaler
S, W, H = int(os.environ.get("SCALE", "2")), int(os.environ.get("W", "1920")), int(os.environ.get("H", "1080"))
n = int(os.environ.get("N", "30")); rounds = int(os.environ.get("ROUNDS", "5"))
w = synth.make_weights(S)
p, b = ncnn_io.build_param_text(S).encode(), ncnn_io.build_bin(w)
src = torch.from_numpy(synth.noise_frame(0, W, H)).cuda()
dst = torch.empty((H * S, W * S, 3), dtype=torch.uint8, device="cuda")
ups, libs = {}, {}
def make(name, path, fused):
    _lib._lib = None
    _lib.LIB_PATH = os.path.abspath(path)
    up = Upscaler(S, param=p, bin=b, tile=int(os.environ.get("TILE", "0")))
    up.set_option("fuse_pairs", fused)
    up.set_option("winograd", int(os.environ.get("WINO", "0")) if fused else 0)
    for _ in range(3):
        up.upscale_device(src.data_ptr(), W, H, dst.data_ptr())
    up.sync()
    up.set_profiling(True)
    ups[name], libs[name] = up, _lib._lib
first = True
for arg in sys.argv[1:]:
    name, path = arg.split("=", 1)
    if first:
        make("unfused", path, 0)
        first = False
    make(name, path, int(os.environ.get("FUSE", "1")))
names = list(ups)
res = {k: [] for k in names}
last = {}
for r in range(rounds):
    for k in (names if r % 2 == 0 else names[::-1]):
        up = ups[k]
        up.reset_stats()
        for _ in range(n):
            up.upscale_device(src.data_ptr(), W, H, dst.data_ptr())
        up.sync()
        st = up.stats()
        res[k].append(st["body_ms_total"] / max(st["body_launches"], 1) * 1e3)
        last.setdefault(k, []).append(st["last_ms_total"] / max(st["frames_timed"], 1) * 1e3)
for k in names:
    v = sorted(res[k])
    lv = sorted(last[k])
    line = f"{k:16s} body per layer median {v[len(v) // 2]:7.2f} us (min {v[0]:7.2f}, max {v[-1]:7.2f}), conv_last {lv[len(lv) // 2]:6.2f} us"
    lib = libs[k]
    if k != "unfused" and hasattr(lib, "reve_debug_read_stamps_pair"):
        for _ in range(n):
            ups[k].upscale_device(src.data_ptr(), W, H, dst.data_ptr())
        ups[k].sync()
        buf = (C.c_ulonglong * (1024 * 16))()
        lib.reve_debug_read_stamps_pair(buf, 1024 * 16)
        a = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 16).astype(np.float64)
        role = (np.arange(1024) % 4) // 2
        ok = a[:, 1] > 0
        wall = (a[:, 3] - a[:, 2]) / 100.0
        clk = (a[:, 5] - a[:, 4]) / np.maximum(wall, 1e-9)
        line += (f" | in-kernel {wall[ok].mean():6.1f} us (slowest wave {wall[ok].max():6.1f}), clock {np.median(clk[ok]):5.0f} MHz ({clk[ok].min():.0f}-{clk[ok].max():.0f}), "
                 f"{a[ok, 1].mean():8.0f} cycles per launch; barrier wait A waves {100 * a[ok & (role == 0), 0].sum() / a[ok & (role == 0), 1].sum():4.1f} %, "
                 f"B waves {100 * a[ok & (role == 1), 0].sum() / a[ok & (role == 1), 1].sum():4.1f} %; cycles per active step (two rows, 4608 of MFMA issue): "
                 f"A {a[ok & (role == 0), 7].sum() / a[ok & (role == 0), 8].sum():6.0f}, B {a[ok & (role == 1), 7].sum() / a[ok & (role == 1), 8].sum():6.0f}; "
                 f"active steps per wave {a[ok, 8].mean():.1f}; cycles before the first step {a[ok, 9].mean():.0f}")
        xcc = a[:, 6].astype(int) & 15
        per = []
        for x in range(8):
            m = ok & (xcc == x)
            if m.any():
                per.append(f"{x}: {wall[m].mean():5.1f} us @ {np.median(clk[m]):4.0f} MHz, blocks {sorted(set((np.nonzero(m)[0] // 4) % 8))}")
        line += "\n    per XCD (HW_REG_XCC_ID: in-kernel time, clock, blockIdx % 8 seen there): " + "; ".join(per)
    print(line, flush=True)
