"""Per-segment cycle totals of k_last in a -DSTAMPS build (scripts/ablate.sh stamps "-DSTAMPS"; REVE_HIP_LIB=reve_amd/abl_stamps.so):
barrier wait, tile set-up, k-loop (with the previous tile's post-process when pipelined), post-process, vmcnt wait.  --x4 for the x4 graph."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from reve_amd import synth, ncnn_io, _lib
from reve_amd.upscaler import Upscaler
W, H = 1920, 1080
S = 4 if "--x4" in sys.argv else (3 if "--x3" in sys.argv else 2)
w = synth.make_weights(S)
up = Upscaler(S, param=ncnn_io.build_param_text(S).encode(), bin=ncnn_io.build_bin(w))
src = torch.from_numpy(synth.noise_frame(0, W, H)).cuda()
dst = torch.empty((H * S, W * S, 3), dtype=torch.uint8, device="cuda")
for _ in range(5):
    up.upscale_device(src.data_ptr(), W, H, dst.data_ptr())
up.sync()
lib = _lib.load()
NW = 2048
buf = (C.c_ulonglong * (NW * 8))()
lib.reve_debug_read_stamps_last.restype = C.c_int
rc = lib.reve_debug_read_stamps_last(buf, NW * 8)
a = np.frombuffer(buf, dtype=np.uint64).reshape(NW, 8).astype(np.float64)
a = a[a.sum(1) > 0]
tot = a.sum(1)
print("rc", rc, "waves", len(a), "mean cycles per wave", tot.mean())
for i, n in enumerate(["barrier", "setup", "k-loop", "post", "vmcnt", "-", "-", "decode"]):
    print(f"{n:8s} mean {a[:, i].mean():10.0f}  ({100 * a[:, i].sum() / tot.sum():5.1f} %)   per tile {a[:, i].mean() / 16:8.0f}")
