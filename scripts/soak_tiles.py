"""One-off sweep on the GPU: random frame sizes x scales x tile sizes through {one layer per launch, tile conv_last} and through
{fused pairs (whole frame or the canvas of planes), strip conv_last — round 6: on tiled frames too, the strips of the planes' interiors}:
identical bytes?  The direct evaluation on both sides (the default, auto, would choose the Winograd pairs: another sum).  550 cases, ~10 s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reve_amd import synth, ncnn_io
from reve_amd.upscaler import Upscaler
rng = np.random.default_rng(7)
bad = 0; n = 0
t0 = time.time()
for scale in (2, 3, 4):
    wts = synth.make_weights(scale)
    p, b = ncnn_io.build_param_text(scale).encode(), ncnn_io.build_bin(wts)
    for tile in (0, 32, 64, 100, 200):
        with Upscaler(scale, param=p, bin=b, tile=tile) as a, Upscaler(scale, param=p, bin=b, tile=tile) as u:
            a.set_option("winograd", 0); a.set_option("fuse_pairs", 0); a.set_option("strip_last", 0)
            u.set_option("winograd", 0); u.set_option("fuse_pairs", 1); u.set_option("strip_last", 1)
            for k in range(60 if scale == 2 else 25):
                w, h = int(rng.integers(1, 700)), int(rng.integers(1, 500))
                img = synth.noise_frame(k, w, h)
                x, y = a.upscale(img), u.upscale(img)
                n += 1
                if not np.array_equal(x, y):
                    bad += 1
                    print("MISMATCH", scale, tile, w, h, int((x != y).sum()), flush=True)
print(f"{n} random shapes x scales x tiles: {bad} mismatches, {time.time() - t0:.0f} s")
