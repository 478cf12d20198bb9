"""Time series of the XCD balancer: per 16 frames the body time per layer, the per-slot mean workgroup running time (tau, us)
and the shares.  env: W, H, N (frames, 320), BAL (1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reve_amd import synth, ncnn_io
from reve_amd.upscaler import Upscaler
S = 2
W, H = int(os.environ.get("W", "1920")), int(os.environ.get("H", "1080"))
n = int(os.environ.get("N", "320"))
w = synth.make_weights(S)
up = Upscaler(S, param=ncnn_io.build_param_text(S).encode(), bin=ncnn_io.build_bin(w))
up.set_option("fuse_pairs", 1)
up.set_option("xcd_balance", int(os.environ.get("BAL", "1")))
src = torch.from_numpy(synth.noise_frame(0, W, H)).cuda()
dst = torch.empty((H * S, W * S, 3), dtype=torch.uint8, device="cuda")
up.set_profiling(True)
for blk in range(n // 16):
    up.reset_stats()
    for _ in range(16):
        up.upscale_device(src.data_ptr(), W, H, dst.data_ptr())
    up.sync()
    st = up.stats()
    tau = [up.get_option(f"xcd_tau_{x}") / 100.0 for x in range(8)]
    sh = [up.get_option(f"xcd_share_{x}") for x in range(8)]
    print(f"frames {blk * 16:4d}: body {st['body_ms_total'] / max(st['body_launches'], 1) * 1e3:7.2f} us/layer; tau {' '.join(f'{t:6.1f}' for t in tau)} (spread {(max(tau) - min(tau)) / max(max(tau), 1e-9) * 100:4.1f} %); shares {sh}; updates {up.get_option('xcd_balance_updates')}", flush=True)
