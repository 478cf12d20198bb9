"""One-off on the GPU: a 7680x4320 frame x2 with the binary's 200-pixel tiling (858 planes on a 5.4 GB canvas: beyond the pair kernel's 32-bit
offsets, so one layer per launch) — conv_last on the strips of the planes' interiors against the tile kernel (bytes identical?) and the first
tile against the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reve_amd import synth, ncnn_io
from reve_amd.upscaler import Upscaler
w = synth.make_weights(2)
p, b = ncnn_io.build_param_text(2).encode(), ncnn_io.build_bin(w)
img = synth.noise_frame(3, 7680, 4320)
outs = {}
for strip in (1, 0):
    with Upscaler(2, param=p, bin=b, tile=200) as up:
        up.set_option("strip_last", strip)
        t0 = time.time(); outs[strip] = up.upscale(img); dt = time.time() - t0
        print("8K x2 tile 200 strip_last", strip, outs[strip].shape, f"{dt:.2f} s", "layers per launch", up.stats()["body_layers_per_launch"], flush=True)
print("identical:", np.array_equal(outs[0], outs[1]), "distinct levels", len(np.unique(outs[1][::16, ::16])))
# a crop against the oracle: the top-left 420 x 420 of the frame is the 2 x 2 tiles there with their aprons = a self-contained 410 x 410 region
from oracle import ref
crop = np.ascontiguousarray(img[:400, :400])
exp = ref.upscale(w, crop, tile=200, prepad=10)
d = np.abs(outs[1][:400, :400].astype(int) - exp[:400, :400].astype(int))
print("vs oracle on the first tile (its apron inside the crop): max", d.max(), "differing", float((d > 0).mean()))
