"""One-off stress of the raw-frame stream pipeline on the GPU: N 1080p frames over G contexts (all on device 0), graph replay on,
callbacks in order, a sample of outputs compared with a synchronous upscale.  env: N (400), G (2)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from reve_amd import synth, ncnn_io, _lib as L
from reve_amd.upscaler import Upscaler
S, W, H = 2, 1920, 1080
n = int(os.environ.get("N", "400")); G = int(os.environ.get("G", "2"))
w = synth.make_weights(S)
p, b = ncnn_io.build_param_text(S).encode(), ncnn_io.build_bin(w)
ups = [Upscaler(S, param=p, bin=b) for _ in range(G)]
for u in ups:
    u.set_option("graph", 1)
frames = [synth.noise_frame(i, W, H) for i in range(8)]
want = {i: ups[0].upscale(frames[i % 8]) for i in (0, 1, 7, n // 2, n - 1)}
lib = L.load()
order, bad = [], []
def rd(_u, i, ptr):
    C.memmove(ptr, frames[i % 8].ctypes.data, W * H * 3)
    return 0
def wr(_u, i, ptr):
    if i in want:
        out = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(H * S, W * S, 3))
        if not np.array_equal(out, want[i]):
            bad.append(i)
    return 0
def dn(_u, i, _a, _b):
    order.append(i)
hs = (C.c_void_p * G)(*[u._h for u in ups])
t0 = time.time()
rc = lib.reve_upscale_stream_multi(hs, G, n, W, H, L.READ_FRAME_CB(rd), L.WRITE_FRAME_CB(wr), L.PROGRESS_CB(dn), None)
dt = time.time() - t0
print(f"rc {rc}; {n} frames over {G} contexts in {dt:.2f} s = {n / dt:.1f} frames/s; callbacks in order: {order == list(range(n))}; mismatching samples: {bad}")
