"""Deterministic synthetic inputs for the realesr-animevideov3 path.

The reference ships neither model files (reve-gui/.gitignore:27-30) nor test frames
for the upscaler, so benches and parity tests use:

* frame streams  S-noise (seed 0x5EED0001, timing: random data avoids the zero-operand
  clock boost) and S-toon (seed 0x5EED0002, flat regions + hard edges + gradient, hits
  the 0 and 255 clamps);
* S-video (seed 0x5EED0003): what a DECODED H.264 anime frame looks like — S-toon with its hard edges softened over three
  pixels (chroma subsampling, deblocking), an offset of up to +-2 LSB per 8x8 block (quantised DC: the block edges a decoder
  leaves) and +-2 LSB of per-pixel grain.  The content the PNG route of an unmodified reve actually carries: flat regions are no
  longer runs of equal bytes;
* synthetic SRVGGNetCompact weights: conv ~ N(0, (0.9/sqrt(fan_in))^2), bias ~ U(-0.05, 0.05),
  PReLU slope ~ U(0.05, 0.3), seed 0x5EED1000 + layer.

Everything is built on splitmix64 evaluated with numpy uint64 arithmetic, so the streams are
a pure function of (seed, index) and identical on every host.
"""
from __future__ import annotations

import hashlib
import numpy as np

SEED_NOISE = 0x5EED0001
SEED_TOON = 0x5EED0002
SEED_VIDEO = 0x5EED0003
SEED_WEIGHTS = 0x5EED1000
FEAT = 64
N_BODY = 16

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser of (x + golden gamma)."""
    with np.errstate(over="ignore"):
        z = (x.astype(np.uint64) + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def _u01(seed: int, n: int, stream: int = 0) -> np.ndarray:
    idx = np.arange(n, dtype=np.uint64)
    base = np.uint64((seed * 0x100000001B3 + stream * 0x9E3779B1) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        bits = splitmix64(base ^ (idx * np.uint64(0x2545F4914F6CDD1D)))
    return ((bits >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def noise_frame(frame: int, w: int, h: int, seed: int = SEED_NOISE) -> np.ndarray:
    """S-noise: u8 = splitmix64(seed ^ (frame << 40) ^ (pixel_index*3 + c)) >> 56."""
    n = w * h * 3
    idx = np.arange(n, dtype=np.uint64)
    key = np.uint64(seed) ^ (np.uint64(frame) << np.uint64(40)) ^ idx
    return (splitmix64(key) >> np.uint64(56)).astype(np.uint8).reshape(h, w, 3)


def toon_frame(frame: int, w: int, h: int, seed: int = SEED_TOON) -> np.ndarray:
    """S-toon: piecewise-flat cells with hard edges over a low-frequency gradient."""
    cell = max(4, min(w, h) // 8)
    yy, xx = np.mgrid[0:h, 0:w]
    cx = (xx + 3 * frame) // cell
    cy = (yy + 2 * frame) // cell
    out = np.empty((h, w, 3), dtype=np.uint8)
    for c in range(3):
        key = np.uint64(seed) ^ (cx.astype(np.uint64) * np.uint64(0x1F123BB5)) ^ (
            cy.astype(np.uint64) * np.uint64(0x5851F42D)) ^ np.uint64(c * 0x9E37)
        flat = (splitmix64(key) >> np.uint64(56)).astype(np.int32)
        # snap a third of the cells to the extremes so both clamps are exercised
        flat = np.where(flat < 48, 0, np.where(flat > 208, 255, flat))
        grad = ((xx * (17 + 5 * c)) // max(w, 1) + (yy * (11 + 3 * c)) // max(h, 1)) - 14
        edge = np.where(((xx + yy + frame) % (cell * 2)) == 0, -96, 0)
        out[..., c] = np.clip(flat + grad + edge, 0, 255).astype(np.uint8)
    return out


def video_frame(frame: int, w: int, h: int, seed: int = SEED_VIDEO) -> np.ndarray:
    """S-video: S-toon as a decoder hands it back — edges softened, +-2 LSB per 8x8 block, +-2 LSB of per-pixel grain."""
    base = toon_frame(frame, w, h).astype(np.int32)
    # (1, 2, 1) / 4 along both axes, edge pixels replicated: a hard edge becomes a three-pixel ramp
    pad = np.pad(base, ((1, 1), (1, 1), (0, 0)), mode="edge")
    soft = (pad[1:-1, :-2] + 2 * pad[1:-1, 1:-1] + pad[1:-1, 2:] + 2) >> 2
    pad = np.pad(soft, ((1, 1), (0, 0), (0, 0)), mode="edge")
    soft = (pad[:-2] + 2 * pad[1:-1] + pad[2:] + 2) >> 2
    yy, xx = np.mgrid[0:h, 0:w]
    out = np.empty((h, w, 3), dtype=np.uint8)
    for c in range(3):
        bkey = np.uint64(seed) ^ (np.uint64(frame) << np.uint64(40)) ^ ((xx // 8).astype(np.uint64) * np.uint64(0x1F123BB5)) ^ (
            (yy // 8).astype(np.uint64) * np.uint64(0x5851F42D)) ^ np.uint64(c * 0x9E37 + 1)
        block = (splitmix64(bkey) >> np.uint64(56)).astype(np.int32) % 5 - 2
        pkey = np.uint64(seed ^ 0xABCD) ^ (np.uint64(frame) << np.uint64(40)) ^ ((yy * w + xx) * 3 + c).astype(np.uint64)
        grain = (splitmix64(pkey) >> np.uint64(56)).astype(np.int32) % 5 - 2
        out[..., c] = np.clip(soft[..., c] + block + grain, 0, 255).astype(np.uint8)
    return out


def _normal(seed: int, n: int) -> np.ndarray:
    u1 = _u01(seed, n, 1)
    u2 = _u01(seed, n, 2)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def make_weights(scale: int, n_body: int = N_BODY, seed: int = SEED_WEIGHTS, fp16_exact: bool = True) -> dict:
    """Synthetic SRVGGNetCompact parameters in PyTorch/ncnn layouts (OIHW, fp32 arrays).

    With fp16_exact the values are rounded to fp16 first (the real model .bin stores fp16
    weight payloads), so writing them to an ncnn .bin is lossless.
    """
    assert scale in (2, 3, 4)
    co_last = 3 * scale * scale

    def conv(layer, co, ci):
        sigma = 0.9 / np.sqrt(ci * 9.0)
        return (_normal(seed + layer, co * ci * 9) * sigma).astype(np.float32).reshape(co, ci, 3, 3)

    def bias(layer, n):
        return ((_u01(seed + layer, n, 3) - 0.5) * 0.1).astype(np.float32)

    def slope(layer, n):
        return (0.05 + 0.25 * _u01(seed + layer, n, 4)).astype(np.float32)

    w = {
        "scale": scale,
        "n_body": n_body,
        "w_first": conv(0, FEAT, 3), "b_first": bias(0, FEAT), "a_first": slope(0, FEAT),
        "w_body": np.stack([conv(1 + l, FEAT, FEAT) for l in range(n_body)]),
        "b_body": np.stack([bias(1 + l, FEAT) for l in range(n_body)]),
        "a_body": np.stack([slope(1 + l, FEAT) for l in range(n_body)]),
        "w_last": conv(1 + n_body, co_last, FEAT), "b_last": bias(1 + n_body, co_last),
    }
    if fp16_exact:
        for k, v in w.items():
            if isinstance(v, np.ndarray):
                w[k] = v.astype(np.float16).astype(np.float32)
    return w


def weights_sha256(w: dict) -> str:
    hsh = hashlib.sha256()
    for k in ("w_first", "b_first", "a_first", "w_body", "b_body", "a_body", "w_last", "b_last"):
        hsh.update(np.ascontiguousarray(w[k], dtype=np.float32).tobytes())
    return hsh.hexdigest()


# ---- weight statistics beyond the one draw above: every parity figure used to come from ONE synthetic draw
# (sigma = 0.9 / sqrt(fan_in), small biases, slopes in [0.05, 0.3]); trained SRVGG layers have heavier tails, larger biases,
# slopes outside [0, 1] and activations of 10^2..10^3.  Each draw is a dict of knobs for make_weights_draw().
WEIGHT_DRAWS = {
    "gain0.5": dict(gain=0.5),
    "gain0.9_other_seed": dict(gain=0.9),
    "gain1.5": dict(gain=1.5),                                        # activations grow to ~20, most outputs saturate
    "student_t3": dict(tail="t3"),
    "student_t3_gain1.5": dict(tail="t3", gain=1.5),
    "bias_pm1": dict(bias=1.0),
    "slopes_-0.2_1.2": dict(slope=(-0.2, 1.2)),                       # the general PReLU form of the kernels (not max(x, slope x))
    "ramp_up_to_3e2": dict(ramp=[2.2] * 16, last_gain=4e-4),          # activations 2 -> 8 -> 28 -> 97 -> 344 over the body
    "ramp_up_to_5e3": dict(ramp=[2.6] * 16, last_gain=1e-4),          # ... -> 107 -> 718 -> 5,000
    "ramp_up_then_down": dict(ramp=[3.0] * 8 + [0.27] * 8),           # up to 334 after eight layers, back to 0.06
    "first_layer_x4": dict(first_gain=4.0),
    "sparse_70pct_zeros": dict(sparsity=0.7),
    "negative_mean": dict(mean_shift=-0.06),
    "everything": dict(tail="t3", bias=0.5, slope=(-0.2, 1.2), ramp=[2.2] * 10 + [0.5] * 6, mean_shift=-0.02),
    "everything_hot": dict(tail="t3", bias=1.0, slope=(-0.2, 1.2), ramp=[2.4] * 14 + [1.0] * 2, last_gain=2e-4),   # up to ~6,000
}


def make_weights_draw(scale: int, name: str, n_body: int = N_BODY) -> dict:
    """Synthetic parameters of the real architecture with the statistics of WEIGHT_DRAWS[name] (fp16-exact values)."""
    k = WEIGHT_DRAWS[name]
    seed = SEED_WEIGHTS + 7919 * (1 + sorted(WEIGHT_DRAWS).index(name))
    gain, tail = k.get("gain", 0.9), k.get("tail", "normal")
    ramp = k.get("ramp", [gain / 0.9 * 0.9 / 0.9] * n_body) if "ramp" in k else [gain / 0.9] * n_body
    co_last = 3 * scale * scale

    def draw(s, n):
        if tail == "t3":          # Student-t, 3 degrees of freedom, scaled to unit variance (variance of t3 is 3)
            z = _normal(s, n)
            chi = (_normal(s + 101, n) ** 2 + _normal(s + 202, n) ** 2 + _normal(s + 303, n) ** 2) / 3.0
            return z / np.sqrt(chi) / np.sqrt(3.0)
        return _normal(s, n)

    def conv(layer, co, ci, g):
        sigma = 0.9 * g / np.sqrt(ci * 9.0)
        v = draw(seed + layer, co * ci * 9) + k.get("mean_shift", 0.0)
        sp = k.get("sparsity", 0.0)
        if sp > 0:
            v = np.where(_u01(seed + layer, co * ci * 9, 7) < sp, 0.0, v / np.sqrt(1.0 - sp))
        return (v * sigma).astype(np.float32).reshape(co, ci, 3, 3)

    def bias(layer, n):
        return ((_u01(seed + layer, n, 3) - 0.5) * 2.0 * k.get("bias", 0.05)).astype(np.float32)

    def slope(layer, n):
        lo, hi = k.get("slope", (0.05, 0.3))
        return (lo + (hi - lo) * _u01(seed + layer, n, 4)).astype(np.float32)

    first_gain = k.get("first_gain", gain / 0.9 if "gain" in k else 1.0)
    w = {
        "scale": scale,
        "n_body": n_body,
        "w_first": conv(0, FEAT, 3, first_gain), "b_first": bias(0, FEAT), "a_first": slope(0, FEAT),
        "w_body": np.stack([conv(1 + l, FEAT, FEAT, ramp[l]) for l in range(n_body)]),
        "b_body": np.stack([bias(1 + l, FEAT) for l in range(n_body)]),
        "a_body": np.stack([slope(1 + l, FEAT) for l in range(n_body)]),
        "w_last": conv(1 + n_body, co_last, FEAT, k.get("last_gain", 1.0)),
        "b_last": ((_u01(seed + 1 + n_body, co_last, 3) - 0.5) * 0.1).astype(np.float32),
    }
    for key, v in w.items():
        if isinstance(v, np.ndarray):
            w[key] = v.astype(np.float16).astype(np.float32)
    return w
