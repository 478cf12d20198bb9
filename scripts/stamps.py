"""Reads the stamps of a -DSTAMPS build of kernels.hip (scripts/ablate.sh stamps "-DSTAMPS"; REVE_HIP_LIB=reve_amd/abl_stamps.so):
cycles per tile, share spent waiting at the tile barrier, in-kernel shader clock (d s_memtime / d s_memrealtime)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from reve_amd import synth, ncnn_io, _lib
from reve_amd.upscaler import Upscaler
W, H, S = 1920, 1080, int(os.environ.get("SCALE", "2"))   # conv_last on the same pipeline: build with -DSTAMP_KIND=2 / 4 (and SCALE=4)
w = synth.make_weights(S)
up = Upscaler(S, param=ncnn_io.build_param_text(S).encode(), bin=ncnn_io.build_bin(w))
src = torch.from_numpy(synth.noise_frame(0, W, H)).cuda()
dst = torch.empty((H * S, W * S, 3), dtype=torch.uint8, device="cuda")
n = int(os.environ.get("N", "400"))          # >= 2 s of back-to-back launches before the stamps that count (the last launch's)
for _ in range(n):
    up.upscale_device(src.data_ptr(), W, H, dst.data_ptr())
up.sync()
lib = _lib.load()
buf = (C.c_ulonglong * (1024 * 8))()
lib.reve_debug_read_stamps2.restype = C.c_int
rc = lib.reve_debug_read_stamps2(buf, 1024 * 8)
a = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.float64)
a = a[a[:, 1] > 0]
tiles = 4080 / 256.0
wall = (a[:, 3] - a[:, 2]) / 100.0           # us
clk = (a[:, 5] - a[:, 4]) / wall             # MHz
print(f"rc {rc}; waves {len(a)}; in-kernel wall time mean {wall.mean():.2f} us (min {wall.min():.2f}, max {wall.max():.2f})")
print(f"in-kernel shader clock: median {np.median(clk):.0f} MHz (min {clk.min():.0f}, max {clk.max():.0f})")
print(f"tile loop: {a[:, 1].mean():.0f} cycles per wave = {a[:, 1].mean() / tiles:.0f} per tile; MFMA issue 9216 per tile = "
      f"{100 * 9216 * tiles / a[:, 1].mean():.1f} % of the loop; barrier wait {a[:, 0].mean() / tiles:.0f} per tile ({100 * a[:, 0].sum() / a[:, 1].sum():.1f} %)")
for x in range(8):
    m = a[:, 6] == x
    if m.any():
        print(f"  XCC {x}: clock {np.median(clk[m]):.0f} MHz, in-kernel {wall[m].mean():.2f} us, loop {a[m, 1].mean():.0f} cycles")
