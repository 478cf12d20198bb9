"""ctypes binding of libreve_hip.so (C ABI: include/reve_hip.h).

There is NO fallback: if the shared library has not been built (`python -c "import
__graft_entry__ as g; g.build()"` or `make -C reve_amd/csrc`) loading fails loudly, and if no
gfx950 device is visible `reve_create` returns REVE_E_NODEVICE.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("REVE_HIP_LIB") or os.path.join(_HERE, "libreve_hip.so")   # env override: A/B builds

REVE_OK = 0
REVE_E_INVALID, REVE_E_MODEL, REVE_E_NODEVICE, REVE_E_HIP = -1, -2, -3, -4
REVE_E_NOMEM, REVE_E_IO, REVE_E_BUSY, REVE_E_UNSUPPORTED = -5, -6, -7, -8


class ReveConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("scale", C.c_int32), ("device", C.c_int32), ("tile", C.c_int32),
        ("prepad", C.c_int32), ("ring_depth", C.c_int32),
        ("model_dir", C.c_char_p), ("model_name", C.c_char_p),
        ("param_data", C.c_void_p), ("param_len", C.c_size_t),
        ("bin_data", C.c_void_p), ("bin_len", C.c_size_t),
    ]


class ReveStats(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("frames_done", C.c_uint64), ("body_launches", C.c_uint64),
        ("body_ms_total", C.c_double), ("frame_ms_last", C.c_double),
        ("h2d_bytes", C.c_uint64), ("d2h_bytes", C.c_uint64),
        ("compute_units", C.c_int32), ("frame_w", C.c_int32), ("frame_h", C.c_int32),
        ("planes", C.c_int32), ("tiles_per_plane", C.c_int32), ("body_layers_per_launch", C.c_int32),
        ("frames_timed", C.c_uint64), ("first_ms_total", C.c_double), ("last_ms_total", C.c_double),
        ("frame_ms_total", C.c_double),
        ("ring_frames", C.c_uint64), ("h2d_ms_total", C.c_double), ("chain_ms_total", C.c_double),
        ("d2h_ms_total", C.c_double), ("ring_wall_ms", C.c_double),
    ]


PROGRESS_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_char_p, C.c_char_p)
READ_FRAME_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p)
WRITE_FRAME_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p)

# every symbol include/reve_hip.h and include/reve_hip_debug.h (the reve_debug_* test probes) declare: (restype, argtypes)
_SIGS = {
    "reve_abi_version": (C.c_int, []),
    "reve_build_info": (C.c_char_p, []),
    "reve_strerror": (C.c_char_p, [C.c_int]),
    "reve_device_count": (C.c_int, []),
    "reve_resolve_model_name": (C.c_int, [C.c_char_p, C.c_int, C.c_char_p, C.c_size_t]),
    "reve_model_report": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_size_t]),
    "reve_create": (C.c_int, [C.POINTER(ReveConfig), C.POINTER(C.c_void_p)]),
    "reve_create_group": (C.c_int, [C.POINTER(ReveConfig), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    "reve_destroy": (None, [C.c_void_p]),
    "reve_last_error": (C.c_char_p, [C.c_void_p]),
    "reve_upscale_rgb8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_ssize_t, C.c_void_p, C.c_ssize_t]),
    "reve_upscale_rgb8_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_ssize_t, C.c_void_p, C.c_ssize_t]),
    "reve_upscale_rgb8_device_batch": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_ssize_t, C.c_ssize_t]),
    "reve_sync": (C.c_int, [C.c_void_p]),
    "reve_submit": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_int, C.c_ssize_t, C.c_void_p, C.c_ssize_t]),
    "reve_wait": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "reve_alloc_pinned": (C.c_void_p, [C.c_size_t]),
    "reve_free_pinned": (None, [C.c_void_p]),
    "reve_upscale_dir": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, PROGRESS_CB, C.c_void_p]),
    "reve_upscale_dir_multi": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_char_p, C.c_char_p, PROGRESS_CB, C.c_void_p]),
    "reve_upscale_file": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p]),
    "reve_upscale_stream_multi": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, READ_FRAME_CB, WRITE_FRAME_CB, PROGRESS_CB, C.c_void_p]),
    "reve_device_cpulist": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    "reve_bind_thread_to_device": (C.c_int, [C.c_int]),
    "reve_trim": (C.c_size_t, []),
    "reve_png_read": (C.c_int, [C.c_char_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "reve_png_write": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_ssize_t]),
    "reve_free": (None, [C.c_void_p]),
    "reve_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "reve_get_stats": (C.c_int, [C.c_void_p, C.POINTER(ReveStats)]),
    "reve_reset_stats": (C.c_int, [C.c_void_p]),
    "reve_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "reve_get_option": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]),
    "reve_debug_blocked_order": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_uint32)]),
    "reve_debug_geometry": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_longlong)]),
    "reve_debug_wino_ring_offset": (C.c_int, [C.c_int, C.c_int]),
    "reve_debug_run_layers": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_ssize_t, C.c_int, C.c_void_p, C.c_size_t]),
    "reve_debug_frames_per_launch": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "reve_debug_model_conditioning": (C.c_int, [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}
EXPORTED_SYMBOLS = tuple(_SIGS)

_lib = None


def load():
    """Loads libreve_hip.so and types every entry point. Raises if the library is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `make -C reve_amd/csrc` or __graft_entry__.build()); there is no CPU fallback")
        # PyTorch-ROCm bundles its own libamdhip64.so.7; two HIP runtimes in one process leave the
        # second without a GPU.  Importing torch first makes the dynamic linker resolve our
        # NEEDED libamdhip64.so.7 to the copy torch already mapped (same SONAME).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)   # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def build_info() -> dict:
    """reve_build_info() as a dict: {"abi": "7", "arch": "gfx950", "pair_src_sha256": "..."}"""
    return dict(kv.split("=", 1) for kv in load().reve_build_info().decode().split())
