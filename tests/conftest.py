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
    # the oracle's OpenMP team: one thread per CPU the process may really use (a box that shows 256 CPUs may allow 16; a team
    # of 256 would then be throttled as a whole, reve_amd/hostcpus.py)
    from reve_amd.hostcpus import usable_cpus
    os.environ.setdefault("OMP_NUM_THREADS", str(usable_cpus()[0]))


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


# with REVE_MODEL_DIR=<dir holding realesr-animevideov3-x{2,3,4}.param/.bin> every test
# that takes `weights` / `model_bytes` runs on those files (the model reve names at reve-shared/src/lib.rs:140-141)
# instead of the synthetic stream: the library loads the file bytes, the oracle gets what ncnn_io.parse_model reads
# from the same bytes.  Tests tied to the synthetic stream's committed outputs (`golden`) are skipped then.
MODEL_DIR = os.environ.get("REVE_MODEL_DIR") or None


@pytest.fixture(scope="session")
def real_model_dir():
    return MODEL_DIR


@pytest.fixture(scope="session")
def model_bytes():
    from reve_amd import ncnn_io, synth
    cache = {}

    def get(scale):
        if scale not in cache:
            if MODEL_DIR:
                cache[scale] = ncnn_io.read_model_files(MODEL_DIR, f"realesr-animevideov3-x{scale}")
            else:
                cache[scale] = (ncnn_io.build_param_text(scale).encode(), ncnn_io.build_bin(synth.make_weights(scale)))
        return cache[scale]

    return get


@pytest.fixture(scope="session")
def weights(model_bytes):
    from reve_amd import ncnn_io, synth
    cache = {}

    def get(scale):
        if scale not in cache:
            if MODEL_DIR:
                p, b = model_bytes(scale)
                cache[scale] = ncnn_io.parse_model(p.decode(), b)
                assert cache[scale]["scale"] == scale, "model file's PixelShuffle factor does not match its name"
            else:
                cache[scale] = synth.make_weights(scale)
        return cache[scale]

    return get


@pytest.fixture(scope="session")
def golden():
    if MODEL_DIR:
        pytest.skip("golden vectors belong to the synthetic weight stream; REVE_MODEL_DIR is set")
    z = np.load(os.path.join(ROOT, "tests", "golden", "srvgg_golden.npz"))
    meta = json.loads(str(z["meta"]))
    return [dict(m, img=z[f"img_{i}"], out=z[f"out_{i}"]) for i, m in enumerate(meta)]


# Reference-held vectors, the day they exist: scripts/pin_against_binary.py turns a directory of PNGs the ORIGINAL
# realesrgan-ncnn-vulkan wrote (with the real model files) into tests/golden/binary_pins/*.npz — inputs, the binary's outputs, the
# tile size they are consistent with and the sha256 of the model files; data only.  A pin is usable when REVE_MODEL_DIR holds the
# model its digest names (the model itself is never committed).  No pin committed = the state of SURVEY.md §8(c): parity unpinned.
PINS_DIR = os.environ.get("REVE_BINARY_PINS") or os.path.join(ROOT, "tests", "golden", "binary_pins")


def load_binary_pins():
    """[(name, meta, [(img, binary_out), ...])] of every pin whose model is at hand; the reasons for the ones that are not"""
    import glob
    import hashlib
    usable, skipped = [], []
    for path in sorted(glob.glob(os.path.join(PINS_DIR, "*.npz"))):
        z = np.load(path)
        meta = json.loads(str(z["meta"]))
        name = os.path.basename(path)
        if not MODEL_DIR:
            skipped.append(f"{name}: REVE_MODEL_DIR is not set (needs {meta['model']}.param/.bin)")
            continue
        ok = True
        for fn, digest in meta["model_sha256"].items():
            fp = os.path.join(MODEL_DIR, fn)
            if not os.path.exists(fp) or hashlib.sha256(open(fp, "rb").read()).hexdigest() != digest:
                skipped.append(f"{name}: {fn} in REVE_MODEL_DIR is not the file the pin was made with")
                ok = False
                break
        if ok:
            usable.append((name, meta, [(z[f"img_{i}"], z[f"out_{i}"]) for i in range(len(meta["frames"]))]))
    return usable, skipped


@pytest.fixture(scope="session")
def binary_pins():
    usable, skipped = load_binary_pins()
    if not usable:
        pytest.skip("no usable pin of the original binary's output (parity unpinned, SURVEY.md §8c): " + ("; ".join(skipped) or f"{PINS_DIR} holds none"))
    return usable


# Full-frame parity figures (max LSB error, histogram) per BASELINE config, written at session end to
# $REVE_PARITY_REPORT (default gpurun_out/parity_report.json: what comes back from the GPU box); the copy the docs
# cite lives under profiles/.
_PARITY = {}


@pytest.fixture(scope="session")
def parity_report():
    def add(name, out, exp, **extra):
        d = np.abs(out.astype(np.int16) - exp.astype(np.int16))
        hist = np.bincount(d.reshape(-1), minlength=3)
        _PARITY[name] = dict(extra, samples=int(d.size), max_lsb=int(d.max()), differing=int(d.size - hist[0]),
                             differing_fraction=float((d.size - hist[0]) / d.size),
                             histogram={str(i): int(n) for i, n in enumerate(hist) if n})
        return _PARITY[name]

    return add


def pytest_sessionfinish(session, exitstatus):
    if not _PARITY:
        return
    path = os.environ.get("REVE_PARITY_REPORT") or os.path.join(ROOT, "gpurun_out", "parity_report.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump({"tolerance_lsb": 1, "model": MODEL_DIR or "synthetic (reve_amd/synth.py)", "cases": _PARITY}, f, indent=1, sort_keys=True)
    except OSError:
        pass


from tests._evaluations import EVALUATIONS, pin_evaluation  # noqa: E402,F401


@pytest.fixture(scope="session")
def upscalers(model_bytes):
    """Cache of GPU contexts keyed by (scale, tile, evaluation)."""
    from reve_amd.upscaler import Upscaler
    cache = {}

    def get(scale, tile=0, evaluation=None, **kw):
        key = (scale, tile, evaluation, tuple(sorted(kw.items())))
        if key not in cache:
            p, b = model_bytes(scale)
            cache[key] = pin_evaluation(Upscaler(scale, param=p, bin=b, tile=tile, **kw), evaluation)
        return cache[key]

    yield get
    for u in cache.values():
        u.close()


def lsb_report(a, b):
    d = np.abs(a.astype(np.int32) - b.astype(np.int32))
    return int(d.max()), float((d > 0).mean())
