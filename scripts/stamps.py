"""Reads the per-segment cycle totals of a -DSTAMPS build (scripts/ablate.sh stamps "-DSTAMPS")."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from reve_amd import synth, ncnn_io, _lib
from reve_amd.upscaler import Upscaler
W, H = 1920, 1080
S = 4 if "--x4" in sys.argv else 2
w = synth.make_weights(S)
up = Upscaler(S, param=ncnn_io.build_param_text(S).encode(), bin=ncnn_io.build_bin(w))
src = torch.from_numpy(synth.noise_frame(0, W, H)).cuda()
dst = torch.empty((H * S, W * S, 3), dtype=torch.uint8, device="cuda")
for _ in range(5):
    up.upscale_device(src.data_ptr(), W, H, dst.data_ptr())
up.sync()
lib = _lib.load()
LAST = "--last" in sys.argv   # conv_last (k_last) instead of the body kernel
NW = 2048 if LAST else 1024
buf = (C.c_ulonglong * (NW * 8))()
rc = (lib.reve_debug_read_stamps_last if LAST else lib.reve_debug_read_stamps)(buf, NW * 8)
a = np.frombuffer(buf, dtype=np.uint64).reshape(NW, 8).astype(np.float64)
a = a[a.sum(1) > 0]
names = (["barrier", "setup", "k-loop", "post", "vmcnt", "-", "-", "decode"] if LAST
         else ["barrier", "setup", "sub0", "sub1", "vmcnt", None, None, "prologue"])
if not LAST:   # columns 5, 6: 100 MHz wall clock at kernel entry / exit
    t0, t1 = a[:, 5].copy(), a[:, 6].copy()
    a[:, 5] = a[:, 6] = 0
    base = t0.min()
    print(f"wall clock (us): entry skew {(t0.max() - base) / 100:.2f}, first exit {(t1.min() - base) / 100:.2f}, "
          f"last exit {(t1.max() - base) / 100:.2f}, mean in-kernel {(t1 - t0).mean() / 100:.2f}")
    if "--xcd" in sys.argv:   # per-XCD (blockIdx % 8) in-kernel time and exit time of wave 0 of each workgroup
        wg = np.arange(len(t0)) // 4
        for x in range(8):
            m = (wg % 8 == x) & (np.arange(len(t0)) % 4 == 0)
            print(f"  XCD {x}: in-kernel mean {(t1[m] - t0[m]).mean() / 100:7.2f} us  min {(t1[m] - t0[m]).min() / 100:7.2f}  "
                  f"max {(t1[m] - t0[m]).max() / 100:7.2f}   prologue {a[m, 7].mean():7.0f} cyc   tiles/loop cycles {a[m, :5].sum(1).mean():9.0f}")
    cyc = a.sum(1).mean() / ((t1 - t0).mean() / 100)
    print(f"shader cycles per us (clock, MHz): {cyc:.0f}")
if not LAST and hasattr(lib, "reve_debug_read_prologue"):
    pb = (C.c_ulonglong * (1024 * 4))()
    lib.reve_debug_read_prologue(pb, 1024 * 4)
    pr = np.frombuffer(pb, dtype=np.uint64).reshape(1024, 4).astype(np.float64)
    print("prologue, cycles since kernel entry: lane constants %.0f, first DMA issued %.0f, weights+tile in LDS %.0f, loop entry %.0f"
          % (pr[:, 0].mean(), pr[:, 1].mean(), pr[:, 2].mean(), a[:, 7].mean()))
tot = a.sum(1)
print("rc", rc, "waves", (tot > 0).sum(), "mean cycles per wave (last launch = conv_last or body?)", tot.mean())
for i, n in enumerate(names):
    if n is None:
        continue
    print(f"{n:8s} mean {a[:, i].mean():10.0f}  ({100 * a[:, i].sum() / tot.sum():5.1f} %)   per tile {a[:, i].mean() / 16:8.0f}")
