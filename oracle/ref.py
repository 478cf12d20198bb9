"""ctypes binding of the CPU oracle (oracle/srvgg_ref.c).

TEST INFRASTRUCTURE ONLY — parity unpinned (see the header of srvgg_ref.c).  May be
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the
product package `reve_amd`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libsrvgg_ref.so")
_lib = None

MODE_FP32 = 0
MODE_FP16_STORAGE = 1
MODE_FP16_WINOGRAD23 = 2     # fp16 storage, 64->64 layers by Winograd F(2x2,3x3): a what-if, not a parity target
MODE_FP16_WINOGRAD43 = 3     # ... F(4x4,3x3)
MODE_FP16_WINOGRAD_ROW = 4   # fp16 storage, body layers by F(2,3) along the row: the HIP path's optional Winograd kernel, restated


class _Weights(C.Structure):
    _fields_ = [("scale", C.c_int), ("n_body", C.c_int)] + [
        (n, C.POINTER(C.c_float))
        for n in ("w_first", "b_first", "a_first", "w_body", "b_body", "a_body", "w_last", "b_last")
    ]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "srvgg_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libsrvgg_ref.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.srvgg_ref_upscale.restype = C.c_int
        _lib.srvgg_ref_upscale.argtypes = [C.POINTER(_Weights), C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_long,
                                           C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int]
        _lib.srvgg_ref_layer.restype = C.c_int
        _lib.srvgg_ref_layer.argtypes = [C.POINTER(_Weights), C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_long,
                                         C.c_int, C.c_void_p]
        _lib.srvgg_f32_to_f16.restype = C.c_uint16
        _lib.srvgg_f32_to_f16.argtypes = [C.c_float]
        _lib.srvgg_f16_to_f32.restype = C.c_float
        _lib.srvgg_f16_to_f32.argtypes = [C.c_uint16]
        _lib.srvgg_ref_num_threads.restype = C.c_int
    return _lib


def _pack(w: dict):
    keep = {}
    s = _Weights()
    s.scale, s.n_body = int(w["scale"]), int(w["n_body"])
    for n in ("w_first", "b_first", "a_first", "w_body", "b_body", "a_body", "w_last", "b_last"):
        a = np.ascontiguousarray(w[n], dtype=np.float32)
        keep[n] = a
        setattr(s, n, a.ctypes.data_as(C.POINTER(C.c_float)))
    return s, keep


def upscale(w: dict, img: np.ndarray, mode: int = MODE_FP16_STORAGE, tile: int = 0, prepad: int = 10,
            nthreads: int = 0) -> np.ndarray:
    """img: HxWx3 uint8 -> (H*s)x(W*s)x3 uint8."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, wd, _ = img.shape
    s = int(w["scale"])
    out = np.empty((h * s, wd * s, 3), dtype=np.uint8)
    ws, keep = _pack(w)
    rc = lib().srvgg_ref_upscale(C.byref(ws), mode, img.ctypes.data, wd, h, wd * 3, out.ctypes.data, wd * s * 3,
                                 tile, prepad, nthreads)
    if rc != 0:
        raise RuntimeError(f"srvgg_ref_upscale failed: {rc}")
    return out


def layer(w: dict, img: np.ndarray, layer_idx: int, mode: int = MODE_FP16_STORAGE) -> np.ndarray:
    """Activation after layer `layer_idx` (0 = conv_first+PReLU, 1..n_body body, n_body+1 = conv_last)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, wd, _ = img.shape
    ch = 64 if layer_idx <= w["n_body"] else 3 * w["scale"] ** 2
    out = np.empty((h, wd, ch), dtype=np.float32)
    ws, keep = _pack(w)
    rc = lib().srvgg_ref_layer(C.byref(ws), mode, img.ctypes.data, wd, h, wd * 3, layer_idx, out.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"srvgg_ref_layer failed: {rc}")
    return out


def num_threads() -> int:
    return lib().srvgg_ref_num_threads()


def alpha_bicubic(a: np.ndarray, scale: int) -> np.ndarray:
    """The alpha plane of an RGBA image, x `scale`, as the binary scales it beside the network's RGB ([UPSTREAM-RECALL]
    realesrgan.cpp: ncnn Interp with resize_type 3 — bicubic, a = -0.75, half-pixel centres, edge samples repeated — on alpha / 255
    stored as fp16, fp32 arithmetic, the result stored as fp16, then clamp(v * 255 + 0.5)).  numpy restatement, written
    independently of reve_amd/csrc/alpha.cpp (matrix form: one weight matrix per axis).  Parity unpinned like the rest."""
    a = np.asarray(a, dtype=np.uint8)
    h, w = a.shape

    def weights(n: int) -> np.ndarray:
        m = np.zeros((n * scale, n), dtype=np.float64)
        A = -0.75
        for d in range(n * scale):
            f = np.float32((d + 0.5) / scale - 0.5)
            s = int(np.floor(f))
            t = np.float32(f - np.float32(s))
            t0, t1, t2 = np.float32(t + 1), t, np.float32(1 - t)
            c0 = np.float32(A * t0 ** 3 - 5 * A * t0 ** 2 + 8 * A * t0 - 4 * A)
            c1 = np.float32((A + 2) * t1 ** 3 - (A + 3) * t1 ** 2 + 1)
            c2 = np.float32((A + 2) * t2 ** 3 - (A + 3) * t2 ** 2 + 1)
            c = [c0, c1, c2, np.float32(np.float32(1) - c0 - c1 - c2)]
            for k in range(4):
                m[d, min(max(s - 1 + k, 0), n - 1)] += float(c[k])
        return m

    x = (a.astype(np.float32) * np.float32(1.0 / 255.0)).astype(np.float16).astype(np.float64)
    y = weights(h) @ (x @ weights(w).T)
    y = y.astype(np.float32).astype(np.float16).astype(np.float32)
    q = y * np.float32(255.0) + np.float32(0.5)
    return np.clip(np.floor(q), 0, 255).astype(np.uint8)
