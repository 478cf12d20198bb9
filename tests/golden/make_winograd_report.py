#!/usr/bin/env python3
"""How far would a Winograd evaluation of the 3x3 layers move the 8-bit output?  

The arithmetic of the reference path lives in realesrgan-ncnn-vulkan + ncnn, which are not available here, so parity with
the real binary is unpinned; the one numeric freedom ncnn's Vulkan path is known to take for 3x3 stride-1 layers with >= 16
channels is a Winograd transform (F(2x2,3x3) or F(4x4,3x3)) instead of the direct sum the oracle (and the HIP kernels)
evaluate.  The oracle's modes 2 / 3 evaluate exactly those layers that way, with fp32 transforms and every intermediate blob
stored as fp16 (oracle/srvgg_ref.c header).  This script reports the LSB histogram of mode 1 (direct, the parity target)
against modes 2 and 3 — and against mode 4, the row-wise F(2,3) evaluation of the HIP path's optional Winograd kernel
(reve_amd/csrc/kernels_wino.hip), body layers only — on the golden inputs and on 1920x1080 frames of both synthetic streams, and writes
tests/golden/winograd_report.json.

Run in the build container:  python tests/golden/make_winograd_report.py [--no-1080p]
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from reve_amd import synth  # noqa: E402
from oracle import ref  # noqa: E402

SMALL = ((48, 40, "toon"), (37, 29, "noise"), (64, 64, "toon"))       # the golden fixtures' inputs (make_golden.py)


def compare(w, img):
    base = ref.upscale(w, img, mode=ref.MODE_FP16_STORAGE)
    out = {"samples": int(base.size)}
    for name, mode in (("winograd_f2x2", ref.MODE_FP16_WINOGRAD23), ("winograd_f4x4", ref.MODE_FP16_WINOGRAD43),
                       ("winograd_row_f23", ref.MODE_FP16_WINOGRAD_ROW)):
        d = np.abs(ref.upscale(w, img, mode=mode).astype(np.int32) - base.astype(np.int32))
        hist = np.bincount(d.ravel(), minlength=2)
        out[name] = {"max_lsb": int(d.max()), "lsb_histogram": [int(x) for x in hist],
                     "fraction_differing": round(float((d > 0).mean()), 6)}
    return out


def small_cases():
    rep = {}
    for scale in (2, 3, 4):
        w = synth.make_weights(scale)
        for wd, ht, kind in SMALL:
            img = synth.toon_frame(1, wd, ht) if kind == "toon" else synth.noise_frame(1, wd, ht)
            rep[f"x{scale}_{wd}x{ht}_{kind}"] = compare(w, img)
    return rep


def main():
    rep = {"what": "u8 output of oracle mode 1 (direct 3x3 sums, fp16 storage: the parity target) against modes 2 / 3 (the same "
                   "layers by Winograd F(2x2,3x3) / F(4x4,3x3), fp32 transforms, fp16 blobs) and mode 4 (body layers by F(2,3) along the row, what "
                   "kernels_wino.hip computes); synthetic weights of the real architecture",
           "small": small_cases()}
    if "--no-1080p" not in sys.argv:
        w = synth.make_weights(2)
        big = {}
        for kind, gen in (("toon", synth.toon_frame), ("noise", synth.noise_frame)):
            t0 = time.time()
            big[f"x2_1920x1080_{kind}"] = compare(w, gen(0, 1920, 1080))
            print(kind, round(time.time() - t0, 1), "s", big[f"x2_1920x1080_{kind}"], flush=True)
        rep["1080p"] = big
    with open(os.path.join(HERE, "winograd_report.json"), "w") as f:
        json.dump(rep, f, indent=1, sort_keys=True)
    print(json.dumps(rep, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
