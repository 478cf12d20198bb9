"""CPU: BASELINE config 1 as the survey meant it — "single 256x256 PNG x2 ... segmentsize=1 (plumbing, no GPU)" — and the multi-GPU
weights broadcast rehearsed without GPUs.

The product has NO CPU compute path and gains none here: `make -C reve_amd/csrc san` links the UNCHANGED host sources
(reve_cli.cpp, main_realesrgan.cpp, capi.cpp, dirmode.cpp, png.cpp, model.cpp, groupcast.cpp, ...) against stand-ins that are
test code (csrc/san/): an engine that upscales by nearest neighbour, a HIP surface of five calls, and a RECORDING table in
place of librccl.  What runs is everything around the arithmetic: argv, model lookup, state files, segment scheduling, the PNG
codec, the `done` protocol, resume (reve-cli/src/main.rs:43-102,249-274,340-343; reve-shared/src/lib.rs:129-155) — under
AddressSanitizer + UBSan — and the one collective of the path against n = 2, 4, 8 fake devices with a failure
injected at every call it makes."""
import json
import os
import subprocess

import numpy as np
import pytest

from reve_amd import ncnn_io, synth
from reve_amd.upscaler import png_read, png_write
from tests.test_sanitizers import ENV, harness, run as run_harness  # noqa: F401  (the fixture builds `make san`)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "reve_amd", "csrc", "build")
STUBS = os.path.join(ROOT, "tests", "stubs")


def nearest(img, s):
    return np.repeat(np.repeat(img, s, axis=0), s, axis=1)


@pytest.fixture()
def models(tmp_path):
    d = tmp_path / "models"
    for s in (2, 3):
        ncnn_io.write_model(str(d), f"realesr-animevideov3-x{s}", synth.make_weights(s))
    return d


def reve(args, cwd, extra=None):
    env = dict(ENV, PATH=STUBS + os.pathsep + os.environ["PATH"], **(extra or {}))
    return subprocess.run([os.path.join(BUILD, "reve_fake")] + args, cwd=cwd, env=env, capture_output=True, text=True, timeout=600)


def test_c1_single_256x256_frame_through_the_executable(tmp_path, harness, models):
    """`realesrgan-hip -i DIR -o DIR -n realesr-animevideov3-x2 -s 2 -f png -v` — the argv of reve-shared/src/lib.rs:134-147 — on
    tmp_frames/0/frame00000001.png: one output file of 512x512, ONE stderr line containing `done` (what reve-cli/src/main.rs:266-273
    counts), nothing else with that substring, exit status 0."""
    src, dst = tmp_path / "tmp_frames" / "0", tmp_path / "out_frames" / "0"
    src.mkdir(parents=True)
    dst.mkdir(parents=True)
    img = synth.toon_frame(0, 256, 256)
    png_write(str(src / "frame00000001.png"), img)
    r = subprocess.run([os.path.join(BUILD, "realesrgan-hip_fake"), "-i", str(src), "-o", str(dst), "-n", "realesr-animevideov3-x2", "-s", "2",
                        "-f", "png", "-v", "-m", str(models)], capture_output=True, text=True, timeout=300, env=ENV)
    assert r.returncode == 0, r.stderr[-2000:]
    done = [l for l in r.stderr.splitlines() if "done" in l]
    assert len(done) == 1 and "frame00000001.png" in done[0], r.stderr
    assert os.listdir(dst) == ["frame00000001.png"]
    out = png_read(str(dst / "frame00000001.png"))
    assert out.shape == (512, 512, 3) and np.array_equal(out, nearest(img, 2))      # (the stand-in engine: nearest neighbour)
    # the GUI's single-file form (reve-gui/src-tauri/src/commands.rs:52-65) and reve-cli's always-x2 name with -s 3 (lib.rs:140-143)
    r = subprocess.run([os.path.join(BUILD, "realesrgan-hip_fake"), "-i", str(src / "frame00000001.png"), "-o", str(tmp_path / "one.png"),
                        "-n", "realesr-animevideov3-x2", "-s", "3", "-m", str(models)], capture_output=True, text=True, timeout=300, env=ENV)
    assert r.returncode == 0, r.stderr[-2000:]
    assert png_read(str(tmp_path / "one.png")).shape == (768, 768, 3)
    # a missing model is an error status and no `done`
    r = subprocess.run([os.path.join(BUILD, "realesrgan-hip_fake"), "-i", str(src), "-o", str(dst), "-n", "nope", "-s", "2", "-m", str(models)],
                       capture_output=True, text=True, timeout=300, env=ENV)
    assert r.returncode != 0 and "done" not in r.stderr


def test_c1_cli_segmentsize_1_state_files_and_resume(tmp_path, harness, models):
    """`reve -i clip.mp4 -s 2 out.mp4 --segmentsize 1` over three 256x256 frames (stub ffmpeg / mediainfo): three segments of one
    frame; the encoder "crashes" on segment 1 -> non-zero exit with args.temp / video.temp kept and segment 0's part done; the
    second run resumes (reve-cli/src/main.rs:43-102), redoes only what is unconfirmed, and the output holds every frame once, in
    order, 512x512."""
    v = tmp_path / "clip.mp4"
    json.dump({"frames": 3, "fps": 24.0, "w": 256, "h": 256}, open(v, "w"))
    out, temp = tmp_path / "out.mp4", tmp_path / "temp"
    base = ["-i", str(v), "--scale", "2", str(out), "--segmentsize", "1", "--temp-dir", str(temp), "--model-dir", str(models)]
    r = reve(base, tmp_path, {"REVE_STUB_FAIL_MERGE": "1"})
    assert r.returncode != 0 and not out.exists(), r.stderr[-2000:]
    args, video = json.loads((temp / "args.temp").read_text()), json.loads((temp / "video.temp").read_text())
    assert set(args) == {"inputpath", "outputpath", "scale", "segmentsize", "crf", "preset", "x265params"} and args["segmentsize"] == 1 and args["scale"] == 2
    assert video["frame_count"] == 3 and video["segment_count"] == 3 and video["segment_size"] == 1 and video["upscale_ratio"] == 2
    assert [s["index"] for s in video["segments"]] == [1, 2] and (temp / "video_parts" / "0.mp4").exists()
    r = reve(["--yes", "--temp-dir", str(temp), "--model-dir", str(models)], tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "resuming upscale" in r.stdout and "done!" in r.stdout and not temp.exists()
    frames = np.load(out)["frames"]
    assert frames.shape == (3, 512, 512, 3)
    for i in range(3):
        assert np.array_equal(frames[i], nearest(synth.toon_frame(i, 256, 256), 2)), i
    # the raw-RGB pipe route through the same scheduler
    out2 = tmp_path / "out2.mp4"
    r = reve(["-i", str(v), "-s", "2", str(out2), "-S", "2", "--temp-dir", str(temp), "--model-dir", str(models), "--io", "pipes"], tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    assert np.array_equal(np.load(out2)["frames"], frames)


@pytest.mark.parametrize("n", [2, 4, 8])
def test_weights_broadcast_sequence_and_unwinding(harness, n):
    """broadcast_blob (groupcast.cpp: what reve_create_group runs for n distinct GPUs) against the recording table: exactly one
    ncclCommInitAll over the n ordinals as given, one stream per device created on that device, ONE GroupStart / GroupEnd around
    n Broadcasts with count = the blob's bytes, ncclUint8, root 0, communicator i on stream i (in place on the root), every
    stream synchronised and destroyed on its device, n CommDestroys; all ranks hold the root's bytes afterwards.  Then a failure
    at CommInitAll, at the first and last stream creation, at GroupStart, at the first and k-th Broadcast, at GroupEnd and at the
    first and last stream sync: an error text comes back and every communicator and stream created is released exactly once
    (use-after-release or a second release would show as BAD in the log; ASan watches the cookies)."""
    for kind in ("asan", "tsan"):
        out = run_harness(harness[kind], "bcast", str(n))
        assert f"bcast: n = {n} ok: {3 + 5 * n} calls in order, {n} ranks hold the blob, 9 injected failures unwound" in out, out


@pytest.mark.parametrize("n", [1, 2, 8])
def test_create_group_over_fake_devices_leaves_nothing_behind(tmp_path, harness, models, n):
    """reve_create_group through the C ABI (capi.cpp, unchanged) over n fake GPUs: n contexts after one broadcast of the 4 KiB
    fake blob; under every injected broadcast failure, with librccl "missing", and with a context that fails to initialise
    half-way, the call returns the documented code (REVE_E_HIP / REVE_E_NOMEM), out[] is all NULL and no engine is alive;
    contexts that share a device (and REVE_GROUP_BCAST=peer) are filled by device-to-device copies without touching RCCL, and
    REVE_GROUP_BCAST=rccl with a device listed twice is REVE_E_INVALID."""
    out = run_harness(harness["asan"], "group", str(n), str(models), "realesr-animevideov3-x2")
    assert f"group: n = {n} ok" in out, out


def test_rgba_image_keeps_its_alpha_channel(tmp_path, harness, models):
    """The GUI's single-file call (reve-gui/src-tauri/src/commands.rs:52-65) with an image that has transparency: the binary
    upscales RGB through the network and the alpha plane beside it by bicubic interpolation; so does `realesrgan-hip -i a.png -o
    b.png` (reve_upscale_file): an RGBA file comes back whose alpha plane is within 1 LSB of the oracle's independent numpy
    restatement (oracle/ref.py: alpha_bicubic), opaque stays opaque, a constant plane stays constant, and 16-bit RGBA input keeps the
    high bytes (what the binary's stb_image does).  On the CPU build the RGB part is the stand-in engine's nearest neighbour."""
    from PIL import Image
    from oracle import ref
    rng = np.random.default_rng(11)
    exe = os.path.join(BUILD, "realesrgan-hip_fake")
    w, h = 37, 23
    rgb = synth.toon_frame(2, w, h)
    yy, xx = np.mgrid[0:h, 0:w]
    planes = {"ramp": ((xx * 255) // (w - 1)).astype(np.uint8), "disc": np.where((xx - 18) ** 2 + (yy - 11) ** 2 < 60, 255, 0).astype(np.uint8),
              "noise": rng.integers(0, 256, (h, w), dtype=np.uint8), "opaque": np.full((h, w), 255, np.uint8), "const": np.full((h, w), 77, np.uint8)}
    for scale in (2, 3):
        for name, a in planes.items():
            src, dst = tmp_path / f"{name}.png", tmp_path / f"{name}_x{scale}.png"
            Image.fromarray(np.dstack([rgb, a])).save(src)
            r = subprocess.run([exe, "-i", str(src), "-o", str(dst), "-n", "realesr-animevideov3", "-s", str(scale), "-m", str(models)],
                               capture_output=True, text=True, timeout=300, env=ENV)
            assert r.returncode == 0, r.stderr[-2000:]
            out = np.array(Image.open(dst))
            assert out.shape == (h * scale, w * scale, 4) and Image.open(dst).mode == "RGBA", (name, out.shape)
            assert np.array_equal(out[..., :3], nearest(rgb, scale))
            exp = ref.alpha_bicubic(a, scale)
            d = np.abs(out[..., 3].astype(int) - exp.astype(int))
            assert d.max() <= 1 and (d > 0).mean() < 0.02, (name, scale, int(d.max()), float((d > 0).mean()))
            if name in ("opaque", "const"):
                assert (out[..., 3] == a[0, 0]).all(), name
    # 16-bit RGBA: high bytes
    a16 = (planes["ramp"].astype(np.uint16) << 8) | 0x5A
    rgba16 = np.dstack([(rgb.astype(np.uint16) << 8) | 0x33, a16])
    import struct, zlib as z
    raw = b"".join(b"\0" + rgba16[y].astype(">u2").tobytes() for y in range(h))

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", z.crc32(t + d))

    (tmp_path / "deep.png").write_bytes(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 6, 0, 0, 0)) + chunk(b"IDAT", z.compress(raw)) + chunk(b"IEND", b""))
    r = subprocess.run([exe, "-i", str(tmp_path / "deep.png"), "-o", str(tmp_path / "deep2.png"), "-s", "2", "-m", str(models)], capture_output=True, text=True, timeout=300, env=ENV)
    assert r.returncode == 0, r.stderr[-2000:]
    out = np.array(Image.open(tmp_path / "deep2.png"))
    assert np.array_equal(out[..., :3], nearest(rgb, 2)) and np.abs(out[..., 3].astype(int) - ref.alpha_bicubic(planes["ramp"], 2).astype(int)).max() <= 1
    # transparency by tRNS — a palette entry, a colour key — is an alpha channel too (the binary's stb_image makes it one)
    pal = Image.fromarray(rgb).quantize(8)
    pal.save(tmp_path / "pal_t.png", transparency=3)
    key = tuple(int(v) for v in rgb[5, 7])
    Image.fromarray(rgb).save(tmp_path / "key_t.png", transparency=key)
    for name, a_in in (("pal_t", np.where(np.array(pal) == 3, 0, 255).astype(np.uint8)), ("key_t", np.where((rgb == np.array(key)).all(-1), 0, 255).astype(np.uint8))):
        assert (a_in == 0).any() and (a_in == 255).any(), name
        r = subprocess.run([exe, "-i", str(tmp_path / f"{name}.png"), "-o", str(tmp_path / f"{name}2.png"), "-s", "2", "-m", str(models)], capture_output=True, text=True, timeout=300, env=ENV)
        assert r.returncode == 0, r.stderr[-2000:]
        out = np.array(Image.open(tmp_path / f"{name}2.png"))
        assert out.shape[2] == 4 and np.abs(out[..., 3].astype(int) - ref.alpha_bicubic(a_in, 2).astype(int)).max() <= 1, name
        assert np.array_equal(out[..., :3], nearest(np.array(Image.open(tmp_path / f"{name}.png").convert("RGB")), 2)), name
    # an opaque file still comes back as plain RGB
    png_write(str(tmp_path / "rgb.png"), rgb)
    r = subprocess.run([exe, "-i", str(tmp_path / "rgb.png"), "-o", str(tmp_path / "rgb2.png"), "-s", "2", "-m", str(models)], capture_output=True, text=True, timeout=300, env=ENV)
    assert r.returncode == 0 and Image.open(tmp_path / "rgb2.png").mode == "RGB"
