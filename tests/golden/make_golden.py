#!/usr/bin/env python3
"""Generates tests/golden/*.npz and cross-checks the C oracle against an independent
torch.nn.functional restatement of SRVGGNetCompact.

Run in the BUILD container (CPU):  python tests/golden/make_golden.py

The reference (ONdraid/reve) holds no golden vector for the upscale path — its only test
checks that out.mp4 exists (reve-cli/tests/run_test.rs:31-34) — and neither the
realesrgan-ncnn-vulkan binary nor its model files are available, so these fixtures pin the
ORACLE'S OWN behaviour (cross-checked against torch, a general library, not reference code).
They do not pin parity with the reference binary: parity stays "unpinned".

A fixture is data only: input frame, weight seed + sha256, expected u8 output.
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from reve_amd import synth  # noqa: E402
from oracle import ref  # noqa: E402


def torch_forward(w, img_u8, h16):
    """Whole-frame forward with torch ops; returns float HxWx3 output blob before post-process."""
    def r(t):
        return t.half().float() if h16 else t

    x = torch.from_numpy(img_u8.astype(np.float32)) * np.float32(1.0 / 255.0)
    x = r(x).permute(2, 0, 1)[None]
    inp = x
    t = lambda a: r(torch.from_numpy(np.asarray(a, dtype=np.float32)))
    x = r(F.conv2d(x, t(w["w_first"]), t(w["b_first"]), padding=1))
    x = r(F.prelu(x, t(w["a_first"])))
    for l in range(w["n_body"]):
        x = r(F.conv2d(x, t(w["w_body"][l]), t(w["b_body"][l]), padding=1))
        x = r(F.prelu(x, t(w["a_body"][l])))
    x = r(F.conv2d(x, t(w["w_last"]), t(w["b_last"]), padding=1))
    x = F.pixel_shuffle(x, w["scale"])
    x = r(x + F.interpolate(inp, scale_factor=w["scale"], mode="nearest"))
    return x[0].permute(1, 2, 0).contiguous().numpy()


def quant(v):
    return np.clip(v * np.float32(255.0) + np.float32(0.5), 0, 255).astype(np.uint8)


def main():
    torch.set_num_threads(8)
    report = {}
    cases = []
    for scale in (2, 3, 4):
        w = synth.make_weights(scale)
        sha = synth.weights_sha256(w)
        for (wd, ht, kind) in ((48, 40, "toon"), (37, 29, "noise"), (64, 64, "toon")):
            img = synth.toon_frame(1, wd, ht) if kind == "toon" else synth.noise_frame(1, wd, ht)
            for mode in (0, 1):
                out = ref.upscale(w, img, mode=mode, tile=0)
                tq = quant(torch_forward(w, img, mode == 1))
                d = np.abs(out.astype(np.int32) - tq.astype(np.int32))
                key = f"x{scale}_{wd}x{ht}_{kind}_m{mode}"
                report[key] = {"max_lsb": int(d.max()), "n_diff": int((d > 0).sum()), "n": int(d.size),
                               "levels": int(len(np.unique(out))), "min": int(out.min()), "max": int(out.max())}
                assert d.max() <= 1, (key, d.max())
                assert (d > 0).mean() < 2e-3, (key, (d > 0).mean())
                for tile in ((0, 32) if mode == 1 else (0,)):
                    o = out if tile == 0 else ref.upscale(w, img, mode=mode, tile=tile, prepad=10)
                    cases.append({"scale": scale, "w": wd, "h": ht, "kind": kind, "mode": mode, "tile": tile,
                                  "weights_sha256": sha, "img": img, "out": o})
    # the synthetic weights must not be degenerate (SURVEY.md §8d acceptance)
    for k, v in report.items():
        assert v["levels"] > 200 or "37x29" in k or "48x40" in k, (k, v)
    np.savez_compressed(
        os.path.join(HERE, "srvgg_golden.npz"),
        meta=json.dumps([{k: v for k, v in c.items() if k not in ("img", "out")} for c in cases]),
        **{f"img_{i}": c["img"] for i, c in enumerate(cases)},
        **{f"out_{i}": c["out"] for i, c in enumerate(cases)},
    )
    with open(os.path.join(HERE, "crosscheck_report.json"), "w") as f:
        json.dump(report, f, indent=1, sort_keys=True)
    print(json.dumps(report, indent=1, sort_keys=True))
    print(f"wrote {len(cases)} cases")


if __name__ == "__main__":
    main()
