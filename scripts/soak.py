"""Race/hazard soak: many distinct frames through the GPU path twice (bitwise equal?) and a sample
against the oracle; sizes chosen to hit partial tiles, multi-round persistent loops and both scales."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reve_amd import synth, ncnn_io
from reve_amd.upscaler import Upscaler
from oracle import ref
n = int(os.environ.get("N", "200"))
bad = 0
for scale, (w, h) in ((2, (1920, 1080)), (2, (333, 777)), (4, (640, 360)), (3, (500, 281))):
    wts = synth.make_weights(scale)
    up = Upscaler(scale, param=ncnn_io.build_param_text(scale).encode(), bin=ncnn_io.build_bin(wts))
    t0 = time.time()
    ref_out = {}
    for i in range(n):
        img = synth.noise_frame(i, w, h) if i % 2 else synth.toon_frame(i, w, h)
        a = up.upscale(img)
        b = up.upscale(img)
        if not np.array_equal(a, b):
            bad += 1
            print("NONDETERMINISTIC", scale, w, h, i, int((a != b).sum()), flush=True)
        if i % 50 == 0:
            y0, x0 = (h // 3) & ~1, (w // 3) & ~1
            crop = img[max(0, y0 - 18):y0 + 40 + 18, max(0, x0 - 18):x0 + 40 + 18]
            exp = ref.upscale(wts, crop)
            oy, ox = (y0 - max(0, y0 - 18)) * scale, (x0 - max(0, x0 - 18)) * scale
            d = np.abs(a[y0 * scale:(y0 + 40) * scale, x0 * scale:(x0 + 40) * scale].astype(int) - exp[oy:oy + 40 * scale, ox:ox + 40 * scale].astype(int))
            if d.max() > 1:
                bad += 1
                print("MISMATCH", scale, w, h, i, int(d.max()), flush=True)
    print(f"x{scale} {w}x{h}: {n} frames x2, {time.time()-t0:.1f}s, bad so far {bad}", flush=True)
    up.close()
print("SOAK", "FAILED" if bad else "OK")
