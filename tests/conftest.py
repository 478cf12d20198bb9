import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build them once, here, with the
    same entry point the driver uses.  On the GPU box the prebuilt files travel with the snapshot."""
    needed = [os.path.join(ROOT, "reve_amd", n) for n in ("libreve_hip.so", "realesrgan-hip", "reve")]
    needed.append(os.path.join(ROOT, "oracle", "libsrvgg_ref.so"))
    if all(os.path.exists(p) for p in needed):
        return
    import shutil
    if shutil.which("hipcc") is None:
        return   # tests that need the library will fail loudly
    import __graft_entry__
    __graft_entry__.build()


def _has_gpu() -> bool:
    try:
        from reve_amd import _lib
        return _lib.load().reve_device_count() > 0
    except Exception:
        return False


HAS_GPU = None


@pytest.fixture(scope="session")
def has_gpu():
    global HAS_GPU
    if HAS_GPU is None:
        HAS_GPU = _has_gpu()
    return HAS_GPU


@pytest.fixture(autouse=True)
def _skip_gpu_without_device(request, has_gpu):
    if request.node.get_closest_marker("gpu") and not has_gpu:
        pytest.skip("no gfx950 device visible")


@pytest.fixture(scope="session")
def weights():
    from reve_amd import synth
    cache = {}

    def get(scale):
        if scale not in cache:
            cache[scale] = synth.make_weights(scale)
        return cache[scale]

    return get


@pytest.fixture(scope="session")
def model_bytes(weights):
    from reve_amd import ncnn_io
    cache = {}

    def get(scale):
        if scale not in cache:
            cache[scale] = (ncnn_io.build_param_text(scale).encode(), ncnn_io.build_bin(weights(scale)))
        return cache[scale]

    return get


@pytest.fixture(scope="session")
def golden():
    z = np.load(os.path.join(ROOT, "tests", "golden", "srvgg_golden.npz"))
    meta = json.loads(str(z["meta"]))
    return [dict(m, img=z[f"img_{i}"], out=z[f"out_{i}"]) for i, m in enumerate(meta)]


@pytest.fixture(scope="session")
def upscalers(model_bytes):
    """Cache of GPU contexts keyed by (scale, tile)."""
    from reve_amd.upscaler import Upscaler
    cache = {}

    def get(scale, tile=0, **kw):
        key = (scale, tile, tuple(sorted(kw.items())))
        if key not in cache:
            p, b = model_bytes(scale)
            cache[key] = Upscaler(scale, param=p, bin=b, tile=tile, **kw)
        return cache[key]

    yield get
    for u in cache.values():
        u.close()


def lsb_report(a, b):
    d = np.abs(a.astype(np.int32) - b.astype(np.int32))
    return int(d.max()), float((d > 0).mean())
