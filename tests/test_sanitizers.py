"""CPU: the host code that parses untrusted files (png.cpp, model.cpp) and runs the directory pipeline's thread
pools (dirmode.cpp: decode pool -> ring -> encode pool, one mutex + condition variable) under AddressSanitizer +
UndefinedBehaviorSanitizer and under ThreadSanitizer (`make -C reve_amd/csrc san`).  The sources are the
product's, unchanged; only the engine behind them is a stand-in (csrc/san/fake_engine.cpp: nearest-neighbour upscale).
GPU sanitizers do not exist on the target pool, so the kernels are outside these builds.

The harness exits 0 unless a sanitizer stops it; rejected inputs are the expected outcome for the corpora."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from reve_amd import ncnn_io, synth
from reve_amd.upscaler import png_write

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "reve_amd", "csrc")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:allocator_may_return_null=1:max_allocation_size_mb=2048",
           UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1")


@pytest.fixture(scope="module")
def harness():
    if not os.path.exists(CLANG):
        pytest.skip("ROCm clang++ (host sanitizer runtimes) not present")
    r = subprocess.run(["make", "-C", CSRC, "san"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return {k: os.path.join(CSRC, "build", f"san_harness_{k}") for k in ("asan", "tsan", "opt")}


def run(exe, *args, timeout=600):
    r = subprocess.run([exe, *args], capture_output=True, text=True, timeout=timeout, env=ENV)
    assert r.returncode == 0, f"{os.path.basename(exe)} {' '.join(args)} -> {r.returncode}\n{r.stdout[-1500:]}\n{r.stderr[-4000:]}"
    return r.stdout


def mutate(data: bytes, rng, n_trunc, n_flip):
    out = []
    for _ in range(n_trunc):
        out.append(data[:int(rng.integers(0, len(data)))])
    for _ in range(n_flip):
        b = bytearray(data)
        for _ in range(int(rng.integers(1, 4))):
            i = int(rng.integers(0, len(b)))
            b[i] ^= 1 << int(rng.integers(0, 8))
        out.append(bytes(b))
    return out


def _png_variants(tmp_path):
    """Valid PNGs of every colour type / depth the decoder accepts (written by Pillow), plus our own encoder's."""
    from PIL import Image
    rng = np.random.default_rng(5)
    files = []
    a = synth.toon_frame(1, 37, 23)
    png_write(str(tmp_path / "own.png"), a)
    files.append(tmp_path / "own.png")
    Image.fromarray(a).save(tmp_path / "rgb.png")
    Image.fromarray(a).convert("L").save(tmp_path / "gray.png")
    Image.fromarray(a).convert("P").save(tmp_path / "pal.png")
    Image.fromarray(a).convert("RGBA").save(tmp_path / "rgba.png")
    Image.fromarray(a).convert("LA").save(tmp_path / "la.png")
    Image.fromarray(a).convert("1").save(tmp_path / "bit.png")
    Image.fromarray((rng.integers(0, 65535, (23, 37))).astype(np.uint16)).save(tmp_path / "g16.png")
    files += [tmp_path / n for n in ("rgb.png", "gray.png", "pal.png", "rgba.png", "la.png", "bit.png", "g16.png")]
    # round 5: Adam7-interlaced files and tRNS transparency (palette entries; a colour key)
    from tests.test_abi import _interlaced_png
    (tmp_path / "adam7.png").write_bytes(_interlaced_png(a, 2))
    (tmp_path / "adam7_rgba.png").write_bytes(_interlaced_png(np.dstack([a, a[..., 0]]), 6))
    Image.fromarray(a).quantize(8).save(tmp_path / "pal_trns.png", transparency=2)
    Image.fromarray(a).save(tmp_path / "rgb_trns.png", transparency=tuple(int(v) for v in a[0, 0]))
    files += [tmp_path / n for n in ("adam7.png", "adam7_rgba.png", "pal_trns.png", "rgb_trns.png")]
    return files


def test_png_corpus_under_asan_ubsan(tmp_path, harness):
    import struct
    import zlib
    (tmp_path / "seed").mkdir()
    seeds = _png_variants(tmp_path / "seed")
    corpus = tmp_path / "png"
    corpus.mkdir()
    rng = np.random.default_rng(11)
    k = 0
    for f in seeds:
        data = f.read_bytes()
        (corpus / f"ok_{f.name}").write_bytes(data)
        for m in mutate(data, rng, 25, 25):
            (corpus / f"m{k:04d}.png").write_bytes(m)
            k += 1

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))

    # well-formed chunks (valid CRCs) with hostile contents: huge dimensions over a tiny IDAT (34 GB claim), wrong
    # inflate size, bad filter byte, palette index past PLTE, zero dimensions, a second IHDR
    sig = b"\x89PNG\r\n\x1a\n"
    def ihdr(w, h, depth=8, ctype=2, il=0):
        return chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, il))
    hostile = {
        "huge": sig + ihdr(65535, 65535, 16, 6) + chunk(b"IDAT", zlib.compress(b"\0" * 64)) + chunk(b"IEND", b""),
        "bomb": sig + ihdr(20000, 20000) + chunk(b"IDAT", zlib.compress(b"\0" * (20001 * 3))) + chunk(b"IEND", b""),
        "short": sig + ihdr(8, 8) + chunk(b"IDAT", zlib.compress(b"\0" * 10)) + chunk(b"IEND", b""),
        "filter9": sig + ihdr(2, 2) + chunk(b"IDAT", zlib.compress(b"\x09" + b"\1" * 6 + b"\x04" + b"\2" * 6)) + chunk(b"IEND", b""),
        "palidx": sig + ihdr(2, 1, 8, 3) + chunk(b"PLTE", b"\1\2\3") + chunk(b"IDAT", zlib.compress(b"\0\0\xff")) + chunk(b"IEND", b""),
        "zero": sig + ihdr(0, 0) + chunk(b"IDAT", zlib.compress(b"")) + chunk(b"IEND", b""),
        "twoihdr": sig + ihdr(2, 2) + ihdr(4000, 4000) + chunk(b"IDAT", zlib.compress(b"\0" * 14)) + chunk(b"IEND", b""),
        "interlaced": sig + ihdr(2, 2, il=1) + chunk(b"IDAT", zlib.compress(b"\0" * 14)) + chunk(b"IEND", b""),
        "paeth": sig + ihdr(3, 3) + chunk(b"IDAT", zlib.compress((b"\x04" + bytes(range(9))) * 3)) + chunk(b"IEND", b""),
        "empty": b"",
    }
    for n, d in hostile.items():
        (corpus / f"h_{n}.png").write_bytes(d)
    out = run(harness["asan"], "png", str(corpus))
    ok, bad = (int(x) for x in (out.split()[1], out.split()[3]))
    assert ok >= len(seeds) + 1 and bad >= 100 and ok + bad == len(seeds) + k + len(hostile), out   # +1: "paeth" is valid


def _fake_cgroup_trees(tmp_path):
    """(root, expected CPUs or None = no limit) for the layouts effective_cpus / usable_cpus read"""
    def tree(name, files):
        root = tmp_path / name
        for rel, text in files.items():
            f = root / rel.lstrip("/")
            f.parent.mkdir(parents=True, exist_ok=True)
            f.write_text(text)
        return str(root)
    return [
        (tree("v2root", {"/sys/fs/cgroup/cpu.max": "200000 100000\n", "/proc/self/cgroup": "0::/\n"}), 2),
        (tree("v2nested", {"/proc/self/cgroup": "12:pids:/x\n0::/a/b\n", "/sys/fs/cgroup/a/cpu.max": "300000 100000\n",
                           "/sys/fs/cgroup/a/b/cpu.max": "max 100000\n"}), 3),
        (tree("v2max", {"/sys/fs/cgroup/cpu.max": "max 100000\n", "/proc/self/cgroup": "0::/\n"}), None),
        (tree("v1", {"/sys/fs/cgroup/cpu/cpu.cfs_quota_us": "150000\n", "/sys/fs/cgroup/cpu/cpu.cfs_period_us": "100000\n"}), 1),
        (tree("v1off", {"/sys/fs/cgroup/cpu/cpu.cfs_quota_us": "-1\n", "/sys/fs/cgroup/cpu/cpu.cfs_period_us": "100000\n"}), None),
        (tree("half", {"/sys/fs/cgroup/cpu.max": "50000 100000\n"}), 1),
        (tree("junk", {"/sys/fs/cgroup/cpu.max": "lots of\n", "/proc/self/cgroup": "0::" + "/x" * 3000 + "\n"}), None),
        (tree("empty", {}), None),
    ]


def test_cpu_budget_reads_affinity_and_cgroup_quota(tmp_path, harness):
    """The codec pools (dirmode.cpp effective_cpus) and the CPU baseline (reve_amd/hostcpus.py) size themselves from the CPUs
    the process may really use: the affinity mask cut down to the control group's quota, cgroup v2 at the mount root or
    along the process's path, cgroup v1, no limit, junk."""
    from reve_amd.hostcpus import usable_cpus
    n = len(os.sched_getaffinity(0))
    for root, want in _fake_cgroup_trees(tmp_path):
        want = n if want is None else min(want, n)
        assert usable_cpus(root) == (want, n), root
        assert run(harness["asan"], "cpus", root).strip() == f"cpus: {want}", root
    assert usable_cpus()[0] >= 1


def test_fast_deflate_under_asan_ubsan(harness):
    """fastdeflate.cpp (directory mode's PNG compressor): 400 synthetic streams of every kind and size, inflated by zlib."""
    out = run(harness["asan"], "deflate", "400", timeout=900)
    assert "400 streams round-tripped" in out


def test_inflate_against_zlib_under_asan_ubsan(harness):
    """The library's own inflate (fastinflate.cpp: what decodes every input frame of directory mode) against zlib's, under ASan +
    UBSan: 500 streams of every block type — stored, fixed, dynamic, with flushes in the middle — written by zlib at levels 0-9 and the
    Z_FIXED / Z_HUFFMAN_ONLY / Z_RLE / Z_FILTERED strategies with random window and memory levels, and by fastdeflate.cpp; each must
    decode to its source, refuse a destination one byte short or long, and never touch the guard byte behind it.  Then 6,000
    damaged copies (truncated anywhere, one to three bits flipped): whatever zlib's uncompress says about a copy — error, or valid
    with these bytes — this decoder must say too."""
    out = run(harness["asan"], "inflate", "500")
    assert "inflate: 500 streams decoded, 6000 damaged copies judged like zlib" in out, out


def test_model_corpus_under_asan_ubsan(tmp_path, harness):
    rng = np.random.default_rng(13)
    corpus = tmp_path / "models"
    corpus.mkdir()
    k = 0
    for scale, fp16 in ((2, True), (3, False), (4, True)):
        w = synth.make_weights(scale)
        p = ncnn_io.build_param_text(scale).encode()
        b = ncnn_io.build_bin(w, fp16=fp16)
        (corpus / f"ok{scale}.param").write_bytes(p)
        (corpus / f"ok{scale}.bin").write_bytes(b)
        for m in mutate(p, rng, 12, 25):       # damaged graph text over the intact weights
            (corpus / f"p{k:04d}.param").write_bytes(m)
            (corpus / f"p{k:04d}.bin").write_bytes(b)
            k += 1
        for m in mutate(b, rng, 12, 6):        # intact graph over damaged weights
            (corpus / f"b{k:04d}.param").write_bytes(p)
            (corpus / f"b{k:04d}.bin").write_bytes(m)
            k += 1
    # hostile sizes in the text: negative and enormous element counts, absurd blob counts
    p2 = ncnn_io.build_param_text(2).encode()
    b2 = (corpus / "ok2.bin").read_bytes()
    for i, (old, new) in enumerate(((b"6=1728", b"6=-1728"), (b"6=1728", b"6=2147483647"), (b"6=36864", b"6=99999999999"),
                                    (b"0=64 1=3", b"0=-64 1=3"), (b" 1 1 ", b" 2147483647 2147483647 "), (b"PReLU", b"PRelu"),
                                    (b"0=64\n", b"0=2147483647\n"), (b"7767517", b"7767517\n" * 3))):
        (corpus / f"h{i}.param").write_bytes(p2.replace(old, new, 1))
        (corpus / f"h{i}.bin").write_bytes(b2)
    (corpus / "nobin.param").write_bytes(p2)
    out = run(harness["asan"], "model", str(corpus))
    ok, bad = (int(x) for x in (out.split()[1], out.split()[3]))
    assert ok >= 3 and bad >= 60, out


@pytest.mark.parametrize("kind,gpus", [("asan", 1), ("tsan", 1), ("tsan", 3)])
def test_directory_pipeline_with_fake_engine(tmp_path, harness, kind, gpus):
    """200 frames through the real dirmode.cpp (thread pools, pinned-buffer pools, ring feeding, in-order callbacks) over
    one and three fake engines; a damaged frame and a frame of another size sit in the middle of the directory."""
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir()
    outd.mkdir()
    for i in range(200):
        w, h = (40, 24) if i != 77 else (24, 40)
        png_write(str(ind / f"frame{i + 1:08d}.png"), synth.toon_frame(i, w, h))
    good = (ind / "frame00000100.png").read_bytes()
    (ind / "frame00000100.png").write_bytes(good[:len(good) // 2])      # truncated: reported as an error, not a crash
    # three engines: the directory is run twice in one process (the second pass re-uses the pinned buffers the first one parked)
    out = run(harness[kind], "dir", str(ind), str(outd), str(gpus), *(["twice"] if gpus == 3 else []), timeout=900)
    assert "in order" in out and "199 outputs checked" in out and "199 callbacks" in out, out
    assert "rc -6" in out and "frame00000100.png" in out                # REVE_E_IO, naming the damaged file
    assert len(os.listdir(outd)) == 199


@pytest.mark.parametrize("kind", ["asan", "tsan"])
def test_raw_frame_stream_with_eight_fake_engines(harness, kind):
    """The pipeline's multi-GPU shape — one feeder thread, ring and pinned pools per engine, shared decode / encode pools,
    callbacks in frame order on the caller's thread — with EIGHT engines and raw frames (no PNG): races and lifetime errors
    would show here."""
    out = run(harness[kind], "stream", "8", "600", "96", "54", timeout=900)
    assert "8 engines, 600 of 600 frames" in out and "in order" in out and "rc 0" in out, out


def test_host_pipeline_capacity_with_eight_engines_that_take_no_time(harness):
    """How many frames per second can the host side push when the GPUs are infinitely fast?  Eight engines whose submit / wait
    cost nothing, 1080p raw frames copied into their pinned buffers by the decode pool (as a decoder would deliver them), the
    sink sampling every page of the 4K result.  The codec pool is the one a 16-CPU budget gives eight GPUs (11 threads: each takes an
    encode job if one is queued, else the next frame to decode; dirmode.cpp) whatever this machine has.  Eight MI355X need 8 x 465 = 3,700 frames/s; the bar is 4,000."""
    from reve_amd.hostcpus import usable_cpus
    env = dict(ENV, REVE_FAKE_ENGINE_NOOP="1", REVE_DIR_THREADS="11")
    best = 0.0
    for _ in range(3):
        r = subprocess.run([harness["opt"], "stream", "8", "6000", "1920", "1080"], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0 and "in order" in r.stdout, r.stdout + r.stderr
        best = max(best, float(r.stdout.split(" = ")[1].split(" frames/s")[0]))
        if best >= 4500:
            break
    print(f"host pipeline capacity, 8 no-op engines, 1080p raw frames: {best:.0f} frames/s on {usable_cpus()[0]} usable CPUs")
    if usable_cpus()[0] >= 8:
        assert best >= 4000, best
    # the same machinery on frames small enough that copying them costs nothing: the lock / queue / callback overhead alone
    r = subprocess.run([harness["opt"], "stream", "8", "40000", "64", "32"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and float(r.stdout.split(" = ")[1].split(" frames/s")[0]) > 20000, r.stdout


def test_gpu_placement_lookup_reads_sysfs(tmp_path, harness):
    """hostbind.cpp: the CPUs and NUMA node next to a GPU come from /sys/bus/pci/devices/<bus id>/{local_cpulist,numa_node};
    the lanes' threads bind to them (intersected with the mask the process already has)."""
    d = tmp_path / "sys" / "bus" / "pci" / "devices" / "0000:c1:00.0"
    d.mkdir(parents=True)
    (d / "local_cpulist").write_text("0-15,128-143\n")
    (d / "numa_node").write_text("1\n")
    out = run(harness["asan"], "cpulist", str(tmp_path), "0000:C1:00.0")      # (HIP prints upper-case hex digits, sysfs lower-case)
    assert "cpulist: '0-15,128-143' (32 CPUs), node 1" in out, out
    out = run(harness["asan"], "cpulist", str(tmp_path), "0000:ff:00.0")      # a box that exposes nothing: empty list, node -1
    assert "cpulist: '' (0 CPUs), node -1" in out, out
