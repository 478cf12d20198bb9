"""Host-side mirror of the reference's upscale interface, over the C ABI.

`Upscaler` is what `Video::upscale_segment` (reve-shared/src/lib.rs:129-155) becomes when the
`realesrgan-ncnn-vulkan` child process is replaced by libreve_hip.so: same inputs (a segment's
frames, the scale, the model) and the same progress contract (one notification per finished
frame, reve-cli/src/main.rs:266-273), but frames are arrays instead of PNG files on disk.
"""
from __future__ import annotations

import ctypes as C
import numpy as np

from . import _lib as L


class ReveError(RuntimeError):
    def __init__(self, code: int, detail: str = ""):
        self.code = code
        text = L.load().reve_strerror(code).decode()
        super().__init__(f"{text} [{code}]" + (f": {detail}" if detail else ""))


def _config(scale, model_dir, model_name, param, bin, device, tile, prepad, ring_depth):
    cfg = L.ReveConfig()
    cfg.struct_size = C.sizeof(L.ReveConfig)
    cfg.scale, cfg.device, cfg.tile, cfg.prepad, cfg.ring_depth = scale, device, tile, prepad, ring_depth
    if param is not None and bin is not None:
        cfg.param_data = C.cast(C.c_char_p(param), C.c_void_p)
        cfg.param_len = len(param)
        cfg.bin_data = C.cast(C.c_char_p(bin), C.c_void_p)
        cfg.bin_len = len(bin)
    else:
        cfg.model_dir = (model_dir or "models").encode()
        cfg.model_name = (model_name or "realesr-animevideov3").encode()
    return cfg


class Upscaler:
    """One reve_ctx (one GPU). scale in {2,3,4}; tile 0 = whole frame, N = ncnn-compat tiling."""

    def __init__(self, scale: int = 2, model_dir: str | None = None, model_name: str | None = None,
                 param: bytes | None = None, bin: bytes | None = None, device: int = 0, tile: int = 0,
                 prepad: int = 10, ring_depth: int = 0, _handle=None):
        lib = L.load()
        self._keep = (param, bin)
        if _handle is None:
            cfg = _config(scale, model_dir, model_name, param, bin, device, tile, prepad, ring_depth)
            h = C.c_void_p()
            rc = lib.reve_create(C.byref(cfg), C.byref(h))
            if rc != 0:
                raise ReveError(rc, lib.reve_last_error(None).decode())
        else:
            h = _handle
        self._h = h
        self._lib = lib
        self.scale = scale
        self._inflight = {}

    def close(self):
        if getattr(self, "_h", None):
            self._lib.reve_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc: int):
        if rc != 0:
            raise ReveError(rc, self._lib.reve_last_error(self._h).decode())

    def upscale(self, frame: np.ndarray) -> np.ndarray:
        """HxWx3 uint8 -> (H*s)x(W*s)x3 uint8, synchronous (reve_upscale_rgb8)."""
        frame = np.ascontiguousarray(frame, dtype=np.uint8)
        h, w, c = frame.shape
        assert c == 3
        out = np.empty((h * self.scale, w * self.scale, 3), dtype=np.uint8)
        self._chk(self._lib.reve_upscale_rgb8(self._h, frame.ctypes.data, w, h, w * 3, out.ctypes.data, w * self.scale * 3))
        return out

    def upscale_device(self, d_src: int, w: int, h: int, d_dst: int, src_stride: int | None = None,
                       dst_stride: int | None = None):
        """Device pointers (e.g. torch tensor.data_ptr()); enqueued on the ctx stream, returns at once."""
        self._chk(self._lib.reve_upscale_rgb8_device(self._h, d_src, w, h, src_stride or w * 3, d_dst,
                                                     dst_stride or w * self.scale * 3))

    def upscale_device_batch(self, d_srcs, d_dsts, w: int, h: int):
        """n frames of one size, device pointers; small frames share their kernel launches (reve_upscale_rgb8_device_batch)."""
        n = len(d_srcs)
        a, b = (C.c_void_p * n)(*d_srcs), (C.c_void_p * n)(*d_dsts)
        self._chk(self._lib.reve_upscale_rgb8_device_batch(self._h, n, a, b, w, h, w * 3, w * self.scale * 3))

    def sync(self):
        self._chk(self._lib.reve_sync(self._h))

    def submit(self, frame_id: int, frame: np.ndarray, out: np.ndarray):
        h, w, _ = frame.shape
        self._inflight[frame_id] = (frame, out)
        self._chk(self._lib.reve_submit(self._h, frame_id, frame.ctypes.data, w, h, frame.strides[0],
                                        out.ctypes.data, out.strides[0]))

    def wait(self) -> int:
        fid = C.c_uint64()
        self._chk(self._lib.reve_wait(self._h, C.byref(fid)))
        self._inflight.pop(fid.value, None)
        return fid.value

    def upscale_segment(self, in_dir: str, out_dir: str, on_done=None) -> int:
        """Directory contract of Video::upscale_segment; on_done(index, in_path, out_path) per frame."""
        n = [0]

        def cb(_user, idx, ip, op):
            n[0] += 1
            if on_done:
                on_done(idx, ip.decode(), op.decode())

        self._chk(self._lib.reve_upscale_dir(self._h, in_dir.encode(), out_dir.encode(), L.PROGRESS_CB(cb), None))
        return n[0]

    def upscale_file(self, in_path: str, out_path: str):
        self._chk(self._lib.reve_upscale_file(self._h, in_path.encode(), out_path.encode()))

    def set_profiling(self, on: bool):
        self._chk(self._lib.reve_set_profiling(self._h, int(on)))

    def set_option(self, name: str, value: int):
        """run-time switch of the context (reve_set_option), e.g. ("fuse_pairs", 1)"""
        self._chk(self._lib.reve_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        v = C.c_int()
        self._chk(self._lib.reve_get_option(self._h, name.encode(), C.byref(v)))
        return v.value

    def stats(self) -> dict:
        s = L.ReveStats()
        s.struct_size = C.sizeof(L.ReveStats)
        self._chk(self._lib.reve_get_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in L.ReveStats._fields_ if k != "struct_size"}

    def reset_stats(self):
        self._chk(self._lib.reve_reset_stats(self._h))

    def debug_layer(self, frame: np.ndarray, layer: int) -> np.ndarray:
        frame = np.ascontiguousarray(frame, dtype=np.uint8)
        h, w, _ = frame.shape
        out = np.empty((h, w, 64 if layer <= 16 else 3 * self.scale ** 2), dtype=np.float32)   # 17 = conv_last (fp16, before PixelShuffle)
        self._chk(self._lib.reve_debug_run_layers(self._h, frame.ctypes.data, w, h, w * 3, layer, out.ctypes.data, out.size))
        return out


class UpscalerGroup:
    """One context per entry of `devices` from a single parse of the model (reve_create_group): the
    weights reach devices[1:] GPU-to-GPU.  Frame f of a segment goes to GPU f mod G;
    `members[g]` is the Upscaler of devices[g]."""

    def __init__(self, devices, scale: int = 2, model_dir: str | None = None, model_name: str | None = None,
                 param: bytes | None = None, bin: bytes | None = None, tile: int = 0, prepad: int = 10,
                 ring_depth: int = 0):
        lib = L.load()
        devices = list(devices)
        cfg = _config(scale, model_dir, model_name, param, bin, devices[0] if devices else 0, tile, prepad, ring_depth)
        n = len(devices)
        hs = (C.c_void_p * max(n, 1))()
        rc = lib.reve_create_group(C.byref(cfg), (C.c_int * max(n, 1))(*devices), n, hs)
        if rc != 0:
            raise ReveError(rc, lib.reve_last_error(None).decode())
        self._lib = lib
        self.scale = scale
        self.members = [Upscaler(scale, param=param, bin=bin, _handle=C.c_void_p(hs[i])) for i in range(n)]

    def close(self):
        for m in self.members:
            m.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def upscale_segment(self, in_dir: str, out_dir: str, on_done=None) -> int:
        """Directory contract of Video::upscale_segment over all GPUs of the group."""
        n = [0]

        def cb(_user, idx, ip, op):
            n[0] += 1
            if on_done:
                on_done(idx, ip.decode(), op.decode())

        hs = (C.c_void_p * len(self.members))(*[m._h for m in self.members])
        rc = self._lib.reve_upscale_dir_multi(hs, len(self.members), in_dir.encode(), out_dir.encode(), L.PROGRESS_CB(cb), None)
        if rc != 0:
            raise ReveError(rc, self._lib.reve_last_error(self.members[0]._h).decode())
        return n[0]


def upscale_stream(members, frames, on_done=None):
    """reve_upscale_stream_multi over the contexts of `members` (Upscaler objects of one scale): `frames` is a list of equal-sized
    HxWx3 uint8 arrays; returns the list of upscaled frames.  Frame f runs on members[f mod len(members)]; the read / write
    callbacks run on the library's pool threads (ctypes takes the GIL for each call)."""
    lib = L.load()
    h, w, _ = frames[0].shape
    s = members[0].scale
    outs = [None] * len(frames)

    def rd(_u, i, p):
        C.memmove(p, np.ascontiguousarray(frames[i]).ctypes.data, w * h * 3)
        return 0

    def wr(_u, i, p):
        outs[i] = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(h * s, w * s, 3)).copy()
        return 0

    def dn(_u, i, _a, _b):
        if on_done:
            on_done(i)

    hs = (C.c_void_p * len(members))(*[m._h for m in members])
    rc = lib.reve_upscale_stream_multi(hs, len(members), len(frames), w, h, L.READ_FRAME_CB(rd), L.WRITE_FRAME_CB(wr), L.PROGRESS_CB(dn), None)
    if rc != 0:
        raise ReveError(rc, lib.reve_last_error(members[0]._h).decode())
    return outs


_PINNED: dict = {}


def pinned_array(shape, dtype=np.uint8) -> np.ndarray:
    """numpy view over hipHostMalloc'ed memory (reve_alloc_pinned). Release with free_pinned()."""
    lib = L.load()
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = lib.reve_alloc_pinned(n)
    if not p:
        raise MemoryError("reve_alloc_pinned failed")
    buf = (C.c_uint8 * n).from_address(p)
    arr = np.frombuffer(buf, dtype=dtype).reshape(shape)
    _PINNED[arr.ctypes.data] = p
    return arr


def free_pinned(arr: np.ndarray):
    p = _PINNED.pop(arr.ctypes.data, None)
    if p:
        L.load().reve_free_pinned(p)


def png_read(path: str) -> np.ndarray:
    """Decodes an 8-bit PNG to HxWx3 uint8 with the library's own codec (no GPU needed)."""
    lib = L.load()
    p = C.POINTER(C.c_uint8)()
    w, h = C.c_int(), C.c_int()
    rc = lib.reve_png_read(path.encode(), C.byref(p), C.byref(w), C.byref(h))
    if rc != 0:
        raise ReveError(rc, lib.reve_last_error(None).decode())
    try:
        return np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()
    finally:
        lib.reve_free(p)


def png_write(path: str, img: np.ndarray):
    lib = L.load()
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w, _ = img.shape
    rc = lib.reve_png_write(path.encode(), img.ctypes.data, w, h, w * 3)
    if rc != 0:
        raise ReveError(rc, lib.reve_last_error(None).decode())
