"""One-off sweep on the GPU for conv_last on tiled frames (round 6, the strip kernel's CANVAS instantiation): random tile sizes x
PREPADS (1 .. 12: the apron the strips may read into shrinks to one pixel) x frame sizes (frames smaller than a tile = one plane with
its apron, edge planes of one pixel) x the x2 / x3 graphs, strips against the tile kernel: identical bytes?  env: CASES (240)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reve_amd import synth, ncnn_io
from reve_amd.upscaler import Upscaler
rng = np.random.default_rng(11)
cases = int(os.environ.get("CASES", "240"))
bad = 0
t0 = time.time()
models = {s: (ncnn_io.build_param_text(s).encode(), ncnn_io.build_bin(synth.make_weights(s))) for s in (2, 3)}
for c in range(cases):
    scale = int(rng.choice([2, 3]))
    tile = int(rng.choice([32, 33, 48, 62, 63, 64, 100, 124, 200, 300]))
    prepad = int(rng.choice([1, 2, 3, 4, 7, 10, 12]))
    w, h = int(rng.integers(1, 520)), int(rng.integers(1, 400))
    if c % 7 == 0:
        w, h = int(rng.integers(1, tile + 1)), int(rng.integers(1, tile + 1))      # one plane
    p, b = models[scale]
    img = synth.noise_frame(c, w, h)
    with Upscaler(scale, param=p, bin=b, tile=tile, prepad=prepad) as s1, Upscaler(scale, param=p, bin=b, tile=tile, prepad=prepad) as s0:
        s0.set_option("strip_last", 0)
        x, y = s0.upscale(img), s1.upscale(img)
    if not np.array_equal(x, y):
        bad += 1
        print("MISMATCH", scale, tile, prepad, w, h, int((x != y).sum()), flush=True)
print(f"{cases} random (tile, prepad, size, scale) cases, strips against the tile kernel: {bad} mismatches, {time.time() - t0:.0f} s")
