"""Does running two frames' kernel chains on two streams (two contexts in one process) beat one stream?
Kernels of different frames are independent, so the second chain can fill the CUs the first one's launch tail,
end-of-kernel L2 write-back and prologue leave idle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reve_amd import synth, ncnn_io
from reve_amd.upscaler import Upscaler
S, W, H = 2, 1920, 1080
n = int(os.environ.get("N", "300")); K = int(os.environ.get("K", "2"))
w = synth.make_weights(S)
p, b = ncnn_io.build_param_text(S).encode(), ncnn_io.build_bin(w)
ups = [Upscaler(S, param=p, bin=b) for _ in range(K)]
src = [torch.from_numpy(synth.noise_frame(i, W, H)).cuda() for i in range(4)]
dst = [torch.empty((H * S, W * S, 3), dtype=torch.uint8, device="cuda") for _ in range(2 * K)]
def run(k, n):
    for i in range(n):
        ups[i % k].upscale_device(src[i % 4].data_ptr(), W, H, dst[i % (2 * k)].data_ptr())
    for u in ups[:k]:
        u.sync()
for k in range(1, K + 1):
    run(k, 20)
for rnd in range(3):
    for k in range(1, K + 1):
        t0 = time.perf_counter(); run(k, n); dt = time.perf_counter() - t0
        print(f"round {rnd}: {k} stream(s): {n / dt:.1f} frames/s", flush=True)
