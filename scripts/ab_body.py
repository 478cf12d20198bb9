"""A/B of body-kernel variants in ONE process on one device, interleaved rounds (cdna_hip_programming.md §5.4 rule 24):
each variant is a context created under its own REVE_BODY value; per round every variant upscales N frames of the same
1080p S-noise input and its body launches are timed by the library's HIP events.  Also checks the outputs bit for bit.

    python scripts/ab_body.py [variants: 1 2 ...]      env: N (frames per round, 40), ROUNDS (5), W, H, SCALE, TILE"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from reve_amd import synth, ncnn_io
from reve_amd.upscaler import Upscaler
variants = sys.argv[1:] or ["1", "2"]
scale = int(os.environ.get("SCALE", "2")); W = int(os.environ.get("W", "1920")); H = int(os.environ.get("H", "1080"))
n = int(os.environ.get("N", "40")); rounds = int(os.environ.get("ROUNDS", "5")); tile = int(os.environ.get("TILE", "0"))
w = synth.make_weights(scale)
p, b = ncnn_io.build_param_text(scale).encode(), ncnn_io.build_bin(w)
ups = {}
for v in variants:
    os.environ["REVE_BODY"] = v
    ups[v] = Upscaler(scale, param=p, bin=b, tile=tile)
src = torch.from_numpy(synth.noise_frame(0, W, H)).cuda()
outs = {}
for v, up in ups.items():
    dst = torch.empty((H * scale, W * scale, 3), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        up.upscale_device(src.data_ptr(), W, H, dst.data_ptr())
    up.sync()
    outs[v] = dst.cpu().numpy()
    up.set_profiling(True)
ref = outs[variants[0]]
for v in variants[1:]:
    d = np.abs(outs[v].astype(int) - ref.astype(int))
    print(f"variant {v} vs {variants[0]}: max diff {d.max()}, differing samples {(d > 0).sum()}", flush=True)
dst = torch.empty((H * scale, W * scale, 3), dtype=torch.uint8, device="cuda")
res = {v: [] for v in variants}
for r in range(rounds):
    for v in (variants if r % 2 == 0 else variants[::-1]):
        up = ups[v]
        up.reset_stats()
        t0 = time.perf_counter()
        for _ in range(n):
            up.upscale_device(src.data_ptr(), W, H, dst.data_ptr())
        up.sync()
        dt = time.perf_counter() - t0
        st = up.stats()
        res[v].append((dt / n * 1e3, st["body_ms_total"] / max(st["body_launches"], 1) * 1e3))
for v in variants:
    body = sorted(x[1] for x in res[v]); fr = sorted(x[0] for x in res[v])
    print(f"REVE_BODY={v}: body launch median {body[len(body) // 2]:.2f} us (min {body[0]:.2f}, max {body[-1]:.2f}); "
          f"frame median {fr[len(fr) // 2]:.4f} ms (min {fr[0]:.4f})", flush=True)
