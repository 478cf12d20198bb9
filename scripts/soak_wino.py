"""One-off sweep on the GPU for the Winograd option on canvases: random frame sizes x tile sizes (the binary's tiling) and random
batches of small frames, every output against the CPU oracle's direct sums with the same tiling (<= 1 LSB, < 1 % of samples) and —
batches — against the same frames one by one (identical bytes).  env: CASES (60), SEED (1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from reve_amd import synth, ncnn_io
from reve_amd.upscaler import Upscaler
from oracle import ref

rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
cases = int(os.environ.get("CASES", "60"))
t0 = time.time()
bad = 0
for scale in (2, 3, 4):
    w = synth.make_weights(scale)
    p, b = ncnn_io.build_param_text(scale).encode(), ncnn_io.build_bin(w)
    ups = {}
    for k in range(cases // 3):
        W, H = int(rng.integers(1, 420)), int(rng.integers(1, 300))
        tile = int(rng.choice([0, 0, 32, 48, 64, 100, 200]))
        if tile not in ups:
            ups[tile] = Upscaler(scale, param=p, bin=b, tile=tile)
            ups[tile].set_option("winograd", 1)
        up = ups[tile]
        img = synth.toon_frame(k, W, H) if k & 1 else synth.noise_frame(k, W, H)
        out = up.upscale(img).astype(np.int32)
        d = np.abs(out - ref.upscale(w, img, tile=tile).astype(np.int32))
        if d.max() > 1 or (d > 0).mean() > 0.012:
            bad += 1
            print(f"MISMATCH x{scale} {W}x{H} tile {tile}: max {d.max()} LSB, {(d > 0).mean():.4f} differ", flush=True)
        if tile == 0 and W * H < 200 * 200:           # a batch of this size: the same bytes as one by one
            n = int(rng.integers(2, 17))
            frames = [synth.noise_frame(1000 + i, W, H) for i in range(n)]
            one = [up.upscale(f) for f in frames]
            src = [torch.from_numpy(f).cuda() for f in frames]
            dst = [torch.empty((scale * H, scale * W, 3), dtype=torch.uint8, device="cuda") for _ in frames]
            up.upscale_device_batch([s.data_ptr() for s in src], [x.data_ptr() for x in dst], W, H)
            up.sync()
            for a_, b_ in zip(dst, one):
                if not np.array_equal(a_.cpu().numpy(), b_):
                    bad += 1
                    print(f"BATCH MISMATCH x{scale} {W}x{H} n {n}", flush=True)
                    break
    for u in ups.values():
        u.close()
print(f"{cases} random shapes x scales x tiles (+ batches) with the Winograd option: {bad} mismatches, {time.time() - t0:.0f} s")
