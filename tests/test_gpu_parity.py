"""GPU parity tests: the HIP path, called through the C ABI (libreve_hip.so), against the CPU
oracle in fp16-storage mode.  Bar: <= 1 LSB per RGB channel (BASELINE.json north_star's stated
tolerance); in practice differences come only from the fp32 summation order inside a conv
(MFMA order vs the oracle's sequential order) flipping an fp16 rounding, so they are rare.

    python -m pytest tests -m gpu -q
"""
import json
import os
import subprocess

import numpy as np
import pytest

from oracle import ref
from reve_amd import synth, ncnn_io
from reve_amd.upscaler import Upscaler, UpscalerGroup, ReveError, pinned_array, free_pinned, png_read, png_write

from tests._evaluations import EVALUATIONS

pytestmark = pytest.mark.gpu

TOL_LSB = 1            # per RGB channel, stated tolerance
MAX_DIFF_FRACTION = 0.01


def check(out, exp, what="", flat=False):
    """flat: a constant-colour input.  Its output is a handful of distinct values repeated over whole regions, so
    one fp16 rounding flipped by the summation order shows up in every pixel of a region: only the +-1 LSB bound
    (the stated tolerance) applies, not the share of differing samples."""
    assert out.shape == exp.shape and out.dtype == np.uint8
    d = np.abs(out.astype(np.int32) - exp.astype(np.int32))
    assert d.max() <= TOL_LSB, f"{what}: max LSB error {d.max()}"
    assert flat or (d > 0).mean() <= MAX_DIFF_FRACTION, f"{what}: {(d > 0).mean():.4%} samples differ"


_RAGGED = {}


def _ragged_oracle(weights, scale, w, h, img):
    if (scale, w, h) not in _RAGGED:          # (one oracle run serves both evaluations)
        _RAGGED[(scale, w, h)] = ref.upscale(weights(scale), img)
    return _RAGGED[(scale, w, h)]


@pytest.mark.parametrize("evaluation", EVALUATIONS)
def test_golden_fixtures(evaluation, golden, upscalers):
    n = 0
    for c in golden:
        if c["mode"] != 1:
            continue   # the GPU implements the reference's fp16-storage numerics only
        up = upscalers(c["scale"], c["tile"], evaluation)
        check(up.upscale(c["img"]), c["out"], f"x{c['scale']} {c['w']}x{c['h']} tile{c['tile']}")
        n += 1
    assert n == 18


@pytest.mark.parametrize("evaluation", EVALUATIONS)
def test_binary_pins(evaluation, binary_pins, model_bytes):
    """The HIP path against outputs of the ORIGINAL binary (tests/golden/binary_pins/, scripts/pin_against_binary.py; consumed when
    REVE_MODEL_DIR holds the model the pin names): <= 1 LSB per RGB sample, north_star's tolerance against the real thing, under both
    evaluations.  Skipped while no pin exists — SURVEY.md §8(c): parity unpinned."""
    for name, meta, frames in binary_pins:
        p, b = model_bytes(meta["scale"])
        with Upscaler(meta["scale"], param=p, bin=b, tile=meta["tile"], prepad=meta["prepad"]) as up:
            up.set_option("winograd", {"direct": 0, "winograd": 1}[evaluation])
            for i, (img, theirs) in enumerate(frames):
                d = np.abs(up.upscale(img).astype(np.int16) - theirs.astype(np.int16))
                assert d.max() <= TOL_LSB, f"{name} frame {i} ({evaluation}): {int(d.max())} LSB from the binary, {float((d > 0).mean()):.3%} of samples differ"


@pytest.mark.parametrize("scale", [2, 3, 4])
def test_layers_against_oracle(scale, upscalers, weights):
    """Kernel-level parity: conv_first and body layers; fp16 activations within 2 ulp-at-2.0.  The direct sums against oracle mode 1
    (the Winograd pairs' activations against their own restatement, mode 4: tests/test_winograd.py)."""
    up = upscalers(scale, 0, "direct")
    w = weights(scale)
    img = synth.toon_frame(1, 70, 45)
    for layer in (0, 1, 2, 3, 7, 15, 16):
        g = up.debug_layer(img, layer)
        o = ref.layer(w, img, layer)
        assert np.isfinite(g).all()
        assert np.abs(g - o).max() <= 2.0 ** -9, f"layer {layer}"
    # first layer: at most a handful of 1-ulp flips
    assert (np.abs(up.debug_layer(img, 0) - ref.layer(w, img, 0)) > 0).mean() < 1e-3
    # conv_last BEFORE PixelShuffle, residual and quantisation (layer 17, the probe instantiation of the conv_last kernel):
    # 12 / 27 / 48 channels of fp16, so a-9 is checked on its own and not only through the u8 output
    g, o = up.debug_layer(img, 17), ref.layer(w, img, 17)
    assert g.shape == o.shape == (45, 70, 3 * scale * scale) and np.isfinite(g).all()
    # (after 17 layers of fp32 sums in another order about four values in ten differ — by an ulp or two of their own fp16 grid)
    assert np.abs(g - o).max() <= 2.0 ** -9 and np.abs(g - o).mean() < 2.0 ** -14, (float(np.abs(g - o).max()), float(np.abs(g - o).mean()))


@pytest.mark.parametrize("scale", [2, 3, 4])
@pytest.mark.parametrize("size", [(1, 1), (2, 3), (17, 5), (32, 16), (33, 17), (31, 15), (65, 33), (100, 100), (129, 50)])
@pytest.mark.parametrize("evaluation", EVALUATIONS)
def test_ragged_sizes(scale, size, evaluation, upscalers, weights):
    """Edge cases: smaller than a tile, exact tile multiples, one past, the reference asset's 100x100."""
    w, h = size
    img = synth.noise_frame(w * 1000 + h, w, h)
    check(upscalers(scale, 0, evaluation).upscale(img), _ragged_oracle(weights, scale, w, h, img), f"x{scale} {w}x{h} {evaluation}")


def test_c1_256x256_x2(upscalers, weights):
    """BASELINE config 1's shape (256x256 -> 512x512, x2)."""
    img = synth.toon_frame(0, 256, 256)
    out = upscalers(2).upscale(img)
    check(out, ref.upscale(weights(2), img), "C1")
    assert len(np.unique(out)) > 200 and out.min() == 0 and out.max() == 255


def test_extreme_inputs(upscalers, weights):
    for img in (np.zeros((40, 50, 3), np.uint8), np.full((40, 50, 3), 255, np.uint8)):
        check(upscalers(2).upscale(img), ref.upscale(weights(2), img), "flat")


def test_prelu_slopes_outside_the_unit_interval(weights):
    """The body kernel uses max(x, slope * x) for a layer whose slopes all lie in [0, 1] (the synthetic model, and what a
    trained PReLU usually holds) and the general form otherwise, chosen per layer at context creation: layers with negative
    slopes, slopes above 1 and exactly 0 / 1 must match the oracle like the others."""
    w = dict(weights(2))
    a = w["a_body"].copy()
    rng = np.random.default_rng(5)
    a[2] = rng.uniform(-0.5, 1.8, 64).astype(np.float16).astype(np.float32)     # general form
    a[5] = np.where(np.arange(64) % 2 == 0, 0.0, 1.0).astype(np.float32)         # the interval's end points: max-form
    a[9, 17] = -0.0625                                                           # one negative slope: general form
    a[12, 3] = 1.0009765625                                                      # one ulp above 1: general form
    w["a_body"] = a
    img = synth.noise_frame(8, 70, 50)
    with Upscaler(2, param=ncnn_io.build_param_text(2).encode(), bin=ncnn_io.build_bin(w)) as up:
        check(up.upscale(img), ref.upscale(w, img), "mixed PReLU forms")
        for layer in (3, 6, 10, 13):     # activations right after the modified layers
            g, o = up.debug_layer(img, layer), ref.layer(w, img, layer)
            assert np.abs(g - o).max() <= 2.0 ** -9 * max(1.0, np.abs(o).max()), layer


@pytest.mark.parametrize("scale", [2, 3, 4])
def test_infinities_clamp_like_the_oracle(scale, weights):
    """fp16 overflow: a conv_last bias beyond 65504 makes every pre-quantisation value +inf / -inf; both sides must clamp to
    255 / 0 (DESIGN.md §3 claims infinities behave alike; NaN, reachable only through inf - inf, is the documented
    exception and is not produced here).  A body-layer bias that overflows exercises inf flowing through PReLU and the
    next convolution's accumulation as well (slopes are positive, so -inf stays -inf)."""
    img = synth.toon_frame(3, 40, 24)
    for sign, where in ((+1, "last"), (-1, "last"), (+1, "body"), (-1, "body")):
        w = dict(weights(scale))
        if where == "last":
            w["b_last"] = np.full_like(w["b_last"], sign * 7.0e4)
        else:
            w["b_body"] = w["b_body"].copy()
            w["b_body"][-1] = sign * 7.0e4                      # last body layer: +-inf activations into conv_last
            w["w_last"] = np.abs(w["w_last"])                   # no inf - inf in conv_last's sums
        exp = ref.upscale(w, img)
        with Upscaler(scale, param=ncnn_io.build_param_text(scale).encode(), bin=ncnn_io.build_bin(w, fp16=False)) as up:
            out = up.upscale(img)
        assert np.array_equal(out, exp), (scale, sign, where, int(np.abs(out.astype(int) - exp.astype(int)).max()))
        assert set(np.unique(exp)) <= {0, 255}


@pytest.mark.parametrize("scale,tile", [(2, 32), (2, 48), (3, 32), (4, 40)])
def test_ncnn_compat_tiles(scale, tile, upscalers, weights):
    """The binary's tiling (N-pixel tiles, 10-px replicate apron, seams and all)."""
    img = synth.toon_frame(3, 90, 70)
    check(upscalers(scale, tile).upscale(img), ref.upscale(weights(scale), img, tile=tile, prepad=10), f"tile{tile}")


def test_tile_200_on_moderate_frame(upscalers, weights):
    img = synth.toon_frame(5, 320, 240)
    check(upscalers(2, 200).upscale(img), ref.upscale(weights(2), img, tile=200, prepad=10), "tile200")


def test_strided_buffers_and_determinism(upscalers, weights):
    up = upscalers(2)
    w, h = 75, 41
    img = synth.noise_frame(9, w, h)
    src = np.zeros((h, w * 3 + 13), np.uint8)
    src[:, :w * 3] = img.reshape(h, -1)
    dst = np.full((h * 2, w * 6 + 7), 0xAB, np.uint8)
    lib = up._lib
    assert lib.reve_upscale_rgb8(up._h, src.ctypes.data, w, h, src.strides[0], dst.ctypes.data, dst.strides[0]) == 0
    a = dst[:, :w * 6].reshape(h * 2, w * 2, 3)
    assert (dst[:, w * 6:] == 0xAB).all(), "wrote past the row"
    check(a, ref.upscale(weights(2), img), "strided")
    assert np.array_equal(a, up.upscale(img)), "not deterministic"


def test_bad_frame_arguments(upscalers):
    up = upscalers(2)
    lib = up._lib
    buf = np.zeros(64, np.uint8)
    assert lib.reve_upscale_rgb8(up._h, buf.ctypes.data, 0, 4, 12, buf.ctypes.data, 24) == -1
    assert lib.reve_upscale_rgb8(up._h, buf.ctypes.data, 4, 4, 3, buf.ctypes.data, 24) == -1   # stride < row
    assert lib.reve_wait(up._h, None) == -7      # nothing in flight
    assert b"nothing in flight" in lib.reve_last_error(up._h)


def test_frame_size_change_reconfigures(upscalers, weights):
    up = upscalers(2)
    for (w, h) in ((40, 30), (90, 20), (40, 30), (16, 64)):
        img = synth.noise_frame(w + h, w, h)
        check(up.upscale(img), ref.upscale(weights(2), img), f"{w}x{h}")
    st = up.stats()
    assert (st["frame_w"], st["frame_h"]) == (16, 64) and st["compute_units"] == 256


def test_async_ring_in_order(model_bytes, weights):
    p, b = model_bytes(2)
    with Upscaler(2, param=p, bin=b, ring_depth=3) as up:
        w, h, n = 64, 48, 7
        frames = [pinned_array((h, w, 3)) for _ in range(3)]
        outs = [pinned_array((h * 2, w * 2, 3)) for _ in range(3)]
        done, expect = [], {}
        for i in range(n):
            if i >= 3:
                fid = up.wait()
                done.append(fid)
                check(outs[fid % 3], expect[fid], f"ring frame {fid}")
            frames[i % 3][...] = synth.noise_frame(i, w, h)
            expect[i] = ref.upscale(weights(2), frames[i % 3])
            up.submit(i, frames[i % 3], outs[i % 3])
        with pytest.raises(ReveError) as e:   # ring full
            up.submit(99, frames[0], outs[0])
        assert e.value.code == -7
        for _ in range(3):
            fid = up.wait()
            done.append(fid)
            check(outs[fid % 3], expect[fid], f"ring frame {fid}")
        assert done == list(range(n))
        st = up.stats()
        assert st["frames_done"] == n and st["h2d_bytes"] == n * w * h * 3 and st["d2h_bytes"] == n * w * h * 12
        for a in frames + outs:
            free_pinned(a)


def test_device_pointer_path(upscalers, weights):
    torch = pytest.importorskip("torch")
    up = upscalers(2)
    w, h = 160, 96
    img = synth.toon_frame(11, w, h)
    src = torch.from_numpy(img).cuda()
    dst = torch.zeros((h * 2, w * 2, 3), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    up.upscale_device(src.data_ptr(), w, h, dst.data_ptr())
    up.sync()
    check(dst.cpu().numpy(), ref.upscale(weights(2), img), "device path")


@pytest.mark.parametrize("scale", [2, 3, 4])
@pytest.mark.parametrize("dst_off,src_off,pad", [(1, 1, 5), (2, 3, 7), (3, 2, 1)])
def test_unaligned_device_pointers_and_strides(scale, dst_off, src_off, pad, upscalers, weights):
    """Frame pointers and row strides with no alignment at all (the x4 kernel stores 4-byte words): the
    bytes around the output rows must stay untouched as well."""
    torch = pytest.importorskip("torch")
    up = upscalers(scale)
    w, h = 53, 37
    img = synth.noise_frame(21 + scale, w, h)
    ss, ds = w * 3 + pad, w * scale * 3 + pad
    sbuf = torch.zeros(src_off + ss * h + 8, dtype=torch.uint8, device="cuda")
    sview = sbuf[src_off:src_off + ss * h].view(h, ss)
    sview[:, :w * 3] = torch.from_numpy(img.reshape(h, w * 3)).cuda()
    dbuf = torch.full((dst_off + ds * h * scale + 8,), 0xA5, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    up.upscale_device(sbuf.data_ptr() + src_off, w, h, dbuf.data_ptr() + dst_off, src_stride=ss, dst_stride=ds)
    up.sync()
    d = dbuf.cpu().numpy()
    rows = d[dst_off:dst_off + ds * h * scale].reshape(h * scale, ds)
    check(rows[:, :w * scale * 3].reshape(h * scale, w * scale, 3), ref.upscale(weights(scale), img), "unaligned")
    assert (rows[:, w * scale * 3:] == 0xA5).all() and (d[:dst_off] == 0xA5).all() and (d[dst_off + ds * h * scale:] == 0xA5).all()


def test_directory_contract(tmp_path, model_bytes, weights):
    """Video::upscale_segment's file contract (reve-shared/src/lib.rs:130-147) through
    reve_upscale_dir: frame%08d.png in -> same stems out, one callback per frame, name order."""
    ind, outd = tmp_path / "tmp_frames" / "0", tmp_path / "out_frames" / "0"
    ind.mkdir(parents=True)
    outd.mkdir(parents=True)
    frames = {}
    for i in (1, 2, 3):
        frames[i] = synth.toon_frame(i, 52, 38)
        png_write(str(ind / f"frame{i:08d}.png"), frames[i])
    (ind / "notes.txt").write_text("ignored")
    p, b = model_bytes(2)
    seen = []
    with Upscaler(2, param=p, bin=b) as up:
        n = up.upscale_segment(str(ind), str(outd), lambda i, a, o: seen.append((i, os.path.basename(a), os.path.basename(o))))
    assert n == 3 and seen == [(i - 1, f"frame{i:08d}.png", f"frame{i:08d}.png") for i in (1, 2, 3)]
    for i in (1, 2, 3):
        check(png_read(str(outd / f"frame{i:08d}.png")), ref.upscale(weights(2), frames[i]), f"dir frame {i}")


def test_group_of_contexts_shards_a_segment(tmp_path, model_bytes, weights):
    """reve_create_group + reve_upscale_dir_multi with devices [0, 0]: two contexts on the one GPU of the
    test box exercise the multi-GPU path: the second context's weights are copied device-to-device,
    frame f goes to context f mod 2, callbacks still arrive once per frame in name order."""
    p, b = model_bytes(2)
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir()
    outd.mkdir()
    frames = [synth.toon_frame(40 + i, 44 + (i == 5) * 8, 30) for i in range(7)]   # one frame of another size
    for i, f in enumerate(frames):
        png_write(str(ind / f"frame{i + 1:08d}.png"), f)
    seen = []
    with UpscalerGroup([0, 0], 2, param=p, bin=b) as grp:
        assert len(grp.members) == 2
        check(grp.members[1].upscale(frames[0]), ref.upscale(weights(2), frames[0]), "member 1 (cloned weights)")
        n = grp.upscale_segment(str(ind), str(outd), lambda i, a, o: seen.append(i))
        st = [m.stats()["frames_done"] for m in grp.members]
    assert n == 7 and seen == list(range(7))
    for i, f in enumerate(frames):
        check(png_read(str(outd / f"frame{i + 1:08d}.png")), ref.upscale(weights(2), f), f"group frame {i}")
    assert st[0] >= 4 and st[1] >= 3 + 1          # both contexts did their share (+1: the direct call above)
    with pytest.raises(ReveError) as e:
        UpscalerGroup([0, 99], 2, param=p, bin=b)
    assert e.value.code == -3                      # REVE_E_NODEVICE, and the first context was released


def test_executable_argv_and_done_protocol(tmp_path, weights):
    """The process-level contract: argv of lib.rs:134-147, one stderr line containing 'done' per
    frame (reve-cli/src/main.rs:266-273), exit status 0; plus the GUI's single-file form."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "reve_amd", "realesrgan-hip")
    models = tmp_path / "models"
    ncnn_io.write_model(str(models), "realesr-animevideov3-x2", weights(2))
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir()
    outd.mkdir()
    imgs = [synth.toon_frame(i, 40, 24) for i in range(2)]
    for i, im in enumerate(imgs):
        png_write(str(ind / f"frame{i + 1:08d}.png"), im)
    r = subprocess.run([exe, "-i", str(ind), "-o", str(outd), "-n", "realesr-animevideov3-x2", "-s", "2", "-f", "png", "-v",
                        "-m", str(models)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert sum(l.endswith(" done") for l in r.stderr.splitlines()) == 2
    # no -t given = the binary's auto tile size (200): same result as the oracle's tile-200 emulation
    check(png_read(str(outd / "frame00000002.png")), ref.upscale(weights(2), imgs[1], tile=200), "exe")
    # -g 0,0: the binary's multi-GPU list form (two contexts on the one GPU here)
    outd2 = tmp_path / "out2"
    outd2.mkdir()
    r = subprocess.run([exe, "-i", str(ind), "-o", str(outd2), "-n", "realesr-animevideov3-x2", "-s", "2", "-v", "-g", "0,0",
                        "-m", str(models)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert sum(l.endswith(" done") for l in r.stderr.splitlines()) == 2
    check(png_read(str(outd2 / "frame00000001.png")), ref.upscale(weights(2), imgs[0], tile=200), "exe -g 0,0")
    # GUI form: -i file -o file -m models -n realesr-animevideov3-x2 -s 2 (commands.rs:52-65)
    r = subprocess.run([exe, "-i", str(ind / "frame00000001.png"), "-o", str(tmp_path / "single.png"), "-m", str(models),
                        "-n", "realesr-animevideov3-x2", "-s", "2", "-t", "full"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    check(png_read(str(tmp_path / "single.png")), ref.upscale(weights(2), imgs[0]), "exe single")
    # ... with an image that has transparency: RGB through the network, the alpha plane scaled beside it (bicubic, as the binary does)
    from PIL import Image
    yy, xx = np.mgrid[0:24, 0:40]
    a = np.where((xx - 20) ** 2 + (yy - 12) ** 2 < 90, 255, (xx * 6) % 256).astype(np.uint8)
    Image.fromarray(np.dstack([imgs[0], a])).save(tmp_path / "rgba.png")
    r = subprocess.run([exe, "-i", str(tmp_path / "rgba.png"), "-o", str(tmp_path / "rgba2.png"), "-m", str(models), "-n", "realesr-animevideov3-x2",
                        "-s", "2", "-t", "full"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    out = np.array(Image.open(tmp_path / "rgba2.png"))
    assert out.shape == (48, 80, 4)
    check(out[..., :3], ref.upscale(weights(2), imgs[0]), "exe rgba: colour")
    assert np.abs(out[..., 3].astype(int) - ref.alpha_bicubic(a, 2).astype(int)).max() <= 1
    # failure is loud: missing model -> non-zero exit, no 'done'
    r = subprocess.run([exe, "-i", str(ind), "-o", str(outd), "-n", "nope", "-s", "2", "-m", str(models)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and not any(l.endswith(" done") for l in r.stderr.splitlines())


def test_executable_tile_choice_from_the_environment(tmp_path, weights):
    """reve never passes -t (reve-shared/src/lib.rs:134-147), so the executable defaults to the binary's 200-pixel tiles and their
    seams; REVE_TILE=full|N opts out without touching reve's argv.  An explicit -t wins; the note it prints
    must not contain the letters reve's progress counter looks for (reve-cli/src/main.rs:266-273)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "reve_amd", "realesrgan-hip")
    models = tmp_path / "models"
    ncnn_io.write_model(str(models), "realesr-animevideov3-x2", weights(2))
    ind = tmp_path / "in"
    ind.mkdir()
    img = synth.toon_frame(5, 96, 64)
    png_write(str(ind / "frame00000001.png"), img)
    argv = [exe, "-i", str(ind), "-o", None, "-n", "realesr-animevideov3-x2", "-s", "2", "-f", "png", "-v", "-m", str(models)]

    def run(name, env_tile, extra=()):
        outd = tmp_path / name
        outd.mkdir()
        a = list(argv)
        a[4] = str(outd)
        env = dict(os.environ)
        env.pop("REVE_TILE", None)
        if env_tile is not None:
            env["REVE_TILE"] = env_tile
        r = subprocess.run(a + list(extra), capture_output=True, text=True, timeout=120, env=env)
        return r, outd / "frame00000001.png"

    tiled32, whole = ref.upscale(weights(2), img, tile=32), ref.upscale(weights(2), img)
    assert (tiled32 != whole).any()                 # (the two choices are distinguishable on this frame: seams)
    r, out = run("full", "full")
    assert r.returncode == 0, r.stderr
    check(png_read(str(out)), whole, "REVE_TILE=full")
    notes = [l for l in r.stderr.splitlines() if "REVE_TILE" in l]
    assert len(notes) == 1 and "done" not in notes[0]
    assert sum(l.endswith(" done") for l in r.stderr.splitlines()) == 1
    r, out = run("t32", "32")
    assert r.returncode == 0, r.stderr
    check(png_read(str(out)), tiled32, "REVE_TILE=32")
    r, out = run("argv_wins", "full", ("-t", "32"))
    assert r.returncode == 0, r.stderr
    check(png_read(str(out)), tiled32, "-t 32 over REVE_TILE=full")
    assert not any("REVE_TILE" in l for l in r.stderr.splitlines())
    r, out = run("bad", "7")
    assert r.returncode == 2 and not out.exists()


@pytest.mark.parametrize("scale", [3, 4])
def test_executable_with_reve_clis_always_x2_name(scale, tmp_path, weights):
    """Drop-in route A for --scale 3 / 4: an unmodified reve-cli passes `-n realesr-animevideov3-x2 -s <scale>`
    (reve-shared/src/lib.rs:140-143).  The executable must load the x<scale> graph, say so without the substring
    'done' (reve-cli/src/main.rs:266-273 counts such lines), and produce the x<scale> result."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "reve_amd", "realesrgan-hip")
    models = tmp_path / "models"
    for s in (2, scale):
        ncnn_io.write_model(str(models), f"realesr-animevideov3-x{s}", weights(s))
    ind, outd = tmp_path / "tmp_frames" / "0", tmp_path / "out_frames" / "0"
    ind.mkdir(parents=True)
    outd.mkdir(parents=True)
    imgs = [synth.toon_frame(30 + i, 50, 36) for i in range(3)]
    for i, im in enumerate(imgs):
        png_write(str(ind / f"frame{i + 1:08d}.png"), im)
    r = subprocess.run([exe, "-i", str(ind), "-o", str(outd), "-n", "realesr-animevideov3-x2", "-s", str(scale), "-f", "png", "-v",
                        "-m", str(models)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = r.stderr.splitlines()
    assert sum("done" in l for l in lines) == 3 and sum(l.endswith(" done") for l in lines) == 3
    assert any(f"realesr-animevideov3-x{scale}" in l and "done" not in l for l in lines)
    for i, im in enumerate(imgs):
        out = png_read(str(outd / f"frame{i + 1:08d}.png"))
        assert out.shape == (36 * scale, 50 * scale, 3)
        check(out, ref.upscale(weights(scale), im, tile=200), f"exe -n x2 -s {scale}")


def test_list_driven_work_order(tmp_path):
    """The shipped whole-frame path computes its 4x8-blocked tile order in the kernel; REVE_NO_BLOCKED_ORDER=1 drives
    the same frames through the work-list variant (what tiled frames always use): same oracle, same tolerance."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from reve_amd import synth, ncnn_io\n"
        "from reve_amd.upscaler import Upscaler\n"
        "from oracle import ref\n"
        "for scale, tile, (w, h) in ((2, 0, (300, 200)), (2, 0, (97, 45)), (2, 48, (150, 90)), (4, 0, (130, 70))):\n"
        "    wts = synth.make_weights(scale)\n"
        "    img = synth.toon_frame(w + h, w, h)\n"
        "    with Upscaler(scale, param=ncnn_io.build_param_text(scale).encode(), bin=ncnn_io.build_bin(wts), tile=tile) as up:\n"
        "        out = up.upscale(img)\n"
        "    d = np.abs(out.astype(int) - ref.upscale(wts, img, tile=tile).astype(int))\n"
        "    assert d.max() <= 1 and (d > 0).mean() < 0.01, (scale, tile, w, h, int(d.max()), float((d > 0).mean()))\n"
        "print('variant ok')\n" % root)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, REVE_NO_BLOCKED_ORDER="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "variant ok" in r.stdout, r.stderr[-2000:]


def test_partial_tiles_every_row_and_column_count(upscalers, weights):
    """Tiles the plane only partly covers (their out-of-plane pixels are computed and dropped by the stores' bounds check; the
    last row of every wave is finished under the NEXT tile's first row): every count of valid rows 1..16 in the bottom tile
    row and of valid columns 1..32 in the right tile column, whole-frame and as ncnn-compat tiles."""
    up = upscalers(2)
    for h in range(17, 33):                       # second tile row holds 1..16 rows
        img = synth.noise_frame(500 + h, 40, h)
        check(up.upscale(img), ref.upscale(weights(2), img), f"rows {h}")
    for w in list(range(33, 65, 3)) + [48, 49, 64]:   # second tile column holds 1..32 columns
        img = synth.noise_frame(600 + w, w, 20)
        check(up.upscale(img), ref.upscale(weights(2), img), f"cols {w}")
    img = synth.toon_frame(9, 150, 130)
    for tile in (60, 100):                        # planes of 80/50 and 120/70 pixels: partial tiles inside the arena
        check(upscalers(2, tile).upscale(img), ref.upscale(weights(2), img, tile=tile, prepad=10), f"tile {tile}")


def test_group_weights_reach_every_context(model_bytes, weights):
    """reve_create_group's one collective: contexts on DISTINCT devices get the packed weights by an RCCL broadcast; with one
    GPU on the test box REVE_GROUP_BCAST=rccl runs that code (librccl loaded, communicator, broadcast, teardown) for a
    one-device group, and the [0, 0, 0] group uses the device-to-device copy.  Both must produce the oracle's frames."""
    p, b = model_bytes(2)
    img = synth.toon_frame(77, 90, 60)
    exp = ref.upscale(weights(2), img)
    os.environ["REVE_GROUP_BCAST"] = "rccl"
    try:
        with UpscalerGroup([0], 2, param=p, bin=b) as grp:
            check(grp.members[0].upscale(img), exp, "rccl one-device group")
        with pytest.raises(Exception, match="distinct devices"):       # forced RCCL cannot serve two contexts on one GPU: loud, not a peer copy
            UpscalerGroup([0, 0], 2, param=p, bin=b)
    finally:
        del os.environ["REVE_GROUP_BCAST"]
    with UpscalerGroup([0, 0, 0], 2, param=p, bin=b) as grp:
        for i, m in enumerate(grp.members):
            check(m.upscale(img), exp, f"peer copy member {i}")


def test_randomised_shapes_strides_and_tiles(model_bytes, weights):
    """Seeded sweep: random frame sizes (1..160), scales, tile modes, padded row strides, content kinds.
    REVE_SWEEP_N / REVE_SWEEP_MAX / REVE_SWEEP_SEED widen it for a one-off hunt (default 200 cases up to 200 px, about 10 s;
    1200 cases up to 320 px ran clean on the final kernels of round 2)."""
    rng = np.random.default_rng(int(os.environ.get("REVE_SWEEP_SEED", "20261002")))
    hi = int(os.environ.get("REVE_SWEEP_MAX", "200")) + 1
    ups = {}
    try:
        for case in range(int(os.environ.get("REVE_SWEEP_N", "200"))):
            scale = int(rng.choice([2, 3, 4]))
            tile = int(rng.choice([0, 0, 32, 48, 200]))
            w, h = int(rng.integers(1, hi)), int(rng.integers(1, hi))
            kind = int(rng.integers(0, 3))
            img = (synth.noise_frame(case, w, h), synth.toon_frame(case, w, h),
                   np.full((h, w, 3), int(rng.integers(0, 256)), np.uint8))[kind]
            if (scale, tile) not in ups:
                p, b = model_bytes(scale)
                ups[(scale, tile)] = Upscaler(scale, param=p, bin=b, tile=tile)
            up = ups[(scale, tile)]
            sp, dp = int(rng.integers(0, 9)), int(rng.integers(0, 9))
            src = np.zeros((h, w * 3 + sp), np.uint8)
            src[:, :w * 3] = img.reshape(h, -1)
            dst = np.full((h * scale, w * scale * 3 + dp), 0x5A, np.uint8)
            rc = up._lib.reve_upscale_rgb8(up._h, src.ctypes.data, w, h, src.strides[0], dst.ctypes.data, dst.strides[0])
            assert rc == 0, up._lib.reve_last_error(up._h)
            assert (dst[:, w * scale * 3:] == 0x5A).all()
            check(dst[:, :w * scale * 3].reshape(h * scale, w * scale, 3), ref.upscale(weights(scale), img, tile=tile, prepad=10),
                  f"case {case}: x{scale} tile{tile} {w}x{h} kind{kind}", flat=(kind == 2))
    finally:
        for u in ups.values():
            u.close()


def test_fp32_payload_model_and_file_loader(tmp_path, weights):
    """ncnn .bin with raw fp32 weight payloads (tag 0) through the file loader (-m/-n path)."""
    w = weights(2)
    ncnn_io.write_model(str(tmp_path), "realesr-animevideov3-x2", w, fp16=False)
    img = synth.toon_frame(5, 77, 31)
    with Upscaler(2, model_dir=str(tmp_path), model_name="realesr-animevideov3") as up:   # "-x2" is appended
        check(up.upscale(img), ref.upscale(w, img), "fp32 payload")


def _bench(args, env=None, timeout=1200):
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, cwd=root,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_json_contract():
    """bench.py prints ONE JSON line with the fields the driver and the judge read; K steps are timed exactly, each a
    batch of frames sized so that the timed region lasts >= 5 s whatever K is; the other BASELINE configurations ride in
    `configs`, one short leg each; the secondary figures are derived from what the library reports, not typed in."""
    import time
    t0 = time.time()
    d = _bench(["--steps", "20", "--warmup", "3"])
    wall = time.time() - t0
    assert wall < 150, wall            # the driver's default run: everything above within its few minutes
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "frames_per_step", "ms_per_frame", "timed_s",
              "pcie_inclusive_fps", "pcie_ring", "stages_ms", "pipeline_fps", "pipeline", "value_is", "configs", "library"):
        assert k in d, k
    assert d["unit"] == "frames/s" and d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 3
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["metric"] == "upscaled frames/sec 1080p->4K x2 realesr-animevideov3" and d["config"]["workload"].startswith("C2: 1920x1080 -> 3840x2160 x2")
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["timed_s"] >= 4.9 and d["frames_per_step"] >= 50 and d["config"]["frames_per_gpu"] == 20 * d["frames_per_step"]
    assert abs(d["value"] - d["frames_per_step"] * 1e3 / d["ms_per_step"]) / d["value"] < 0.01
    assert abs(d["value"] - 1e3 / d["ms_per_frame"]) / d["value"] < 0.01
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "traffic_stale", "layers_per_launch", "launch_us",
              "algorithmic_flop_per_launch", "mfma_flop_executed", "mfma_instructions_per_launch", "launch_geometry"):
        assert k in rf, k
    # a traffic figure that was not measured in this run says where it comes from, and whether the library timed here was built
    # from the kernel sources it was measured on
    assert (rf["traffic"] is None) == (rf["traffic_source"] is None) == (rf["traffic_stale"] is None)
    tj = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")))
    # (the record of the kernel that ran: the Winograd pairs under the default evaluation)
    assert rf["winograd"] is True and "wino_hbm_bytes_per_launch" in rf["traffic_source"] and rf["traffic"] == tj["wino_hbm_bytes_per_launch"]
    assert rf["traffic_stale"] == (tj.get("wino_src_sha256") != d["library"]["wino_src_sha256"])
    assert 1.0 < rf["traffic"] / (2 * 1920 * 1080 * 128) < 1.08          # no wasted re-reads: the launch's algorithmic bytes + the strips' halo columns
    assert rf["layers_per_launch"] in (1, 2) and rf["algorithmic_flop_per_launch"] == rf["layers_per_launch"] * 2 * 36864 * 1920 * 1080
    assert abs(rf["achieved"] - rf["algorithmic_flop_per_launch"] / (rf["launch_us"] * 1e-6) / 1e12) < 0.01 * rf["achieved"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0.2 < rf["frac"] < 1.0
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    # the headline is the library's default evaluation: option "winograd" = auto, which chooses the Winograd pairs for these weights
    # (kappa 0.017 against the limit 0.5) — the line says which evaluation ran and why
    assert d["roofline"]["frames_per_launch"] == 1 and rf["evaluation"].startswith("winograd F(2,3) along the row (auto: kappa 0.01")
    assert 0.005 < rf["kappa"] < 0.05 and rf["kappa_limit"] == 0.5 and "k_wino" in rf["kernel"]
    # executed MFMA work: derived from the launch geometry — 31 strips x 8 segments (7 of 136 rows, one of 128) on 256 CUs; a unit of
    # NB rows runs ceil((NB + 2) / 2) + ceil(NB / 2) steps of 2 waves x 192 MFMAs (288 for the direct sums): two thirds of the
    # direct kernel's count, 0.69 of the ALGORITHMIC flops the roofline prices
    g = rf["launch_geometry"]
    assert g == {"strips": 31, "segments": 8, "seg_rows": 136, "units": 248}, g
    assert rf["mfma_instructions_per_launch"] == 31 * (7 * (69 + 68) + (65 + 64)) * 2 * 192 == 19427328 * 2 // 3
    assert rf["mfma_flop_executed"] == 19427328 * 2 // 3 * 16384 and 0.68 < rf["mfma_flop_executed"] / rf["algorithmic_flop_per_launch"] < 0.70
    # the informational leg with the direct kernels pinned (REVE_WINOGRAD=0, round 5's default): present at N = 1, never the headline
    assert "option_winograd" not in d
    od = d["option_direct"]
    assert od["unit"] == "frames/s" and od["frames"] > 0 and od["evaluation"] == "direct" and 0.8 * d["value"] < od["value"] < 1.0 * d["value"]
    assert cb["kind"] == "port" and cb["unit"] == "frames/s" and cb["cores"] >= 1 and cb["value"] > 0
    # per-stage times: the frame's kernels, and the three stages of the host ring with its overlap efficiency
    sm = d["stages_ms"]
    assert sm["frames_timed"] > 0 and 0 < sm["conv_first"] < sm["conv_last"] < sm["body_x16"] < sm["chain"]
    assert abs(sm["conv_first"] + sm["body_x16"] + sm["conv_last"] - sm["chain"]) < 0.02 * sm["chain"]
    ring = d["pcie_ring"]
    assert ring["frames"] > 0 and ring["h2d_ms"] > 0 and ring["d2h_ms"] > ring["h2d_ms"] and ring["chain_ms"] > 0
    assert ring["slowest_stage"] in ("h2d", "chain", "d2h") and 0.5 < ring["overlap_efficiency"] <= 1.02
    assert 0.5 * d["value"] < d["pcie_inclusive_fps"] <= 1.02 * d["value"]
    # the pipeline north_star names is measured over the same number of frames as the headline, for >= 5 s, and is a key of its own
    pl = d["pipeline"]
    assert d["pipeline_fps"] == d["pcie_inclusive_fps"] and pl["frames"] == d["config"]["frames_per_gpu"] and pl["timed_s"] >= 4.9
    assert pl["ring_depth"] >= 3 and pl["pcie_bound_fps"] > d["pipeline_fps"]
    # every other BASELINE configuration, timed on this box in this process
    cf = d["configs"]
    assert set(cf) == {"C2_tile200", "C3", "C3_literal", "C5", "C4_1gpu"}
    for name, c in cf.items():
        for k in ("workload", "value", "unit", "frames", "timed_s", "roofline", "launch_us", "pipeline_fps", "pcie_bound_fps", "slowest_stage",
                  "roofline_frac_whole_path", "frames_per_launch"):
            assert k in c, (name, k)
        assert c["unit"] == "frames/s" and c["timed_s"] >= (4.1 if name == "C4_1gpu" else 1.45) and c["frames"] >= 8, (name, c)
        assert 0.2 < c["roofline"]["frac"] < 1.0 and c["launch_us"] == c["roofline"]["launch_us"] > 0, (name, c["roofline"])
        assert c["roofline"]["winograd"] is True and c["roofline"]["evaluation"].startswith("winograd") and c["roofline"]["kappa"] < 0.05, (name, c["roofline"])
        assert 0.5 * c["value"] < c["pipeline_fps"] <= 1.03 * c["value"] and c["pcie_bound_fps"] > 0.95 * c["pipeline_fps"], (name, c)
    assert "200-px tiles" in cf["C2_tile200"]["workload"] and 0.6 * d["value"] < cf["C2_tile200"]["value"] < 0.95 * d["value"]
    assert "7680x4320 x4" in cf["C3"]["workload"] and 0.85 * d["value"] < cf["C3"]["value"] < 1.02 * d["value"]
    assert "3840x2160 x4" in cf["C3_literal"]["workload"] and cf["C3_literal"]["frames_per_launch"] == 4 and cf["C3_literal"]["value"] > 2.5 * d["value"]
    assert "7680x4320 x2" in cf["C5"]["workload"] and 0.2 * d["value"] < cf["C5"]["value"] < 0.3 * d["value"]
    # config 4's stream on the one GPU: whole 1000-frame segments, each completed before the next, at the headline's rate
    c4 = cf["C4_1gpu"]
    assert c4["segmentsize"] == 1000 and c4["segments"] >= 3 and c4["frames"] > 2000 and 0.95 * d["value"] < c4["value"] < 1.03 * d["value"]


def test_launch_geometry_matches_the_pmc_count():
    """The MFMA instructions a body-pair launch executes, as the library derives them from its launch geometry (option
    "pair_mfma_per_launch"), against the matrix pipes' own count: profiles/traffic.json keeps SQ_VALU_MFMA_BUSY_CYCLES / 16 of
    the 1080p launch from the round's rocprofv3 --pmc pass (scripts/install_profiles.py)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tj = json.load(open(os.path.join(root, "profiles", "traffic.json")))
    pmc = tj.get("pair_mfma_instructions_per_launch_pmc")
    if pmc is None:
        rs = json.load(open(os.path.join(root, "profiles", "r04", "pmc_summary.json")))
        pmc = int(round([v for k, v in rs.items() if "k_pair" in k][0]["SQ_VALU_MFMA_BUSY_CYCLES"] / 16))
    from reve_amd import ncnn_io
    w = synth.make_weights(2)
    with Upscaler(2, param=ncnn_io.build_param_text(2).encode(), bin=ncnn_io.build_bin(w)) as up:
        up.set_option("winograd", 0)          # (the count below is the direct kernel's, whatever REVE_WINOGRAD says)
        up.upscale(synth.noise_frame(0, 1920, 1080))
        n = up.get_option("pair_mfma_per_launch")
        assert n == 19427328 and abs(n - pmc) <= 0.002 * n, (n, pmc)
        up.set_option("winograd", 1)
        assert up.get_option("pair_mfma_per_launch") == n * 2 // 3
        up.set_option("winograd", 0)
        up.upscale(synth.noise_frame(0, 960, 540))          # four frames per launch: the stacked canvas' geometry
        assert up.get_option("batch_frames") == 4 and up.get_option("pair_strips") == 16 and up.get_option("pair_units") <= 256


def test_bench_small_frames_share_their_launches():
    """`bench.py --workload WxH`: the sizes of the reference's own assets (reve-cli/assets/: 100x100, 640x480; BASELINE config 1:
    256x256).  Frames that small go through the kernel chain several per launch; the line says how many, prices the launch's
    algorithmic FLOP accordingly and the batched run beats one frame per launch."""
    common = ["--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--min-timed-s", "0.5"]
    d = _bench(common + ["--workload", "256x256"])
    one = _bench(common + ["--workload", "256x256", "--batch", "0"])
    r = d["roofline"]
    assert r["frames_per_launch"] == 16 and one["roofline"]["frames_per_launch"] == 1
    # (executed: the default evaluation, Winograd, runs two thirds of the direct sums' MFMAs — which exceed the algorithmic count by
    # the strips' recomputed columns and the segments' halo rows, here of sixteen small frames)
    assert r["winograd"] is True and r["algorithmic_flop_per_launch"] == 2 * 73728 * 256 * 256 * 16
    assert 1.0 < 1.5 * r["mfma_flop_executed"] / r["algorithmic_flop_per_launch"] < 1.5
    assert d["config"]["frames_per_launch"] == 16 and "256x256" in d["config"]["workload"] and d["frames_per_step"] % 16 == 0
    assert d["value"] > 2.5 * one["value"], (d["value"], one["value"])
    assert d["pipeline"]["ring_depth"] == 32 and d["pipeline_fps"] > 1.5 * one["pipeline_fps"]
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "C9"], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode != 0 and "WxH" in (r.stdout + r.stderr)


def test_bench_launches_its_own_ranks():
    """`bench.py --gpus 2` with no torchrun environment must start two ranks itself (here sharing the one GPU, control
    plane over gloo) and say n_gpus 2; a rank count that disagrees with --gpus must fail instead of printing a line."""
    import sys
    d = _bench(["--gpus", "2", "--steps", "10", "--warmup", "2", "--no-cpu-baseline",
                "--segmentsize", "100", "--min-timed-s", "0.5"], env={"REVE_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["steps"] == 10 and d["scaling"] == "weak"
    # no --workload: with more than one rank the line is BASELINE config 4's schedule (segments, each completed before the next)
    assert d["config"]["workload"].startswith("C4: 1920x1080 -> 3840x2160 x2") and "segments of 100 frames" in d["config"]["workload"]
    assert d["metric"] == "upscaled frames/sec 1080p->4K x2 realesr-animevideov3" and "n1_is" in d["config"]
    assert d["config"]["frames_total"] == 2 * d["config"]["frames_per_gpu"] and d["config"]["segments"] >= 2
    assert "of every segment" in d["config"]["frame_sharding"]
    # what a slow rank would look like from rank 0
    pr = d["per_rank"]
    assert [m["rank"] for m in pr] == [0, 1] and d["slowest_rank"]["rank"] in (0, 1) and 0.5 < d["slowest_rank"]["of_mean"] <= 1.0
    for m in pr:
        for k in ("fps", "pipeline_fps", "bound_cpus", "local_cpulist", "pinned_alloc_ms", "bcast_ms", "device", "backend", "launch_us", "host_pinned_GBps"):
            assert k in m, k
        assert m["fps"] > 0 and m["frames"] == d["config"]["frames_per_gpu"] and m["backend"] == "gloo" and m["bcast_ms"] > 0 and m["pinned_alloc_ms"] > 0
    assert abs(sum(m["fps"] for m in pr) - d["value"]) < 0.25 * d["value"]      # (ranks that share one GPU finish at different times)
    assert d["host_pinned_GBps"] == pytest.approx(d["pipeline_fps"] * 1920 * 1080 * 3 * 5 / 1e9, rel=1e-3)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2"], capture_output=True, text=True,
                       timeout=600, cwd=root, env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stdout + r.stderr) and "{" not in r.stdout
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--steps", "2"], capture_output=True, text=True,
                       timeout=600, cwd=root)   # RCCL backend: three distinct devices needed, one present
    assert r.returncode != 0 and "{" not in r.stdout


def test_two_contexts_from_two_threads(model_bytes, weights):
    """include/reve_hip.h: entry points are re-entrant across different contexts.  Two contexts (x2 and x4) on
    the same GPU are driven from two threads at once; every result must match the oracle."""
    import threading
    jobs = []
    for scale, seed in ((2, 1), (4, 2)):
        p, b = model_bytes(scale)
        imgs = [synth.noise_frame(seed * 100 + i, 150 + 7 * i, 90 + 3 * i) for i in range(6)]
        jobs.append((scale, Upscaler(scale, param=p, bin=b), imgs, [None] * len(imgs)))
    errs = []

    def run(job):
        scale, up, imgs, outs = job
        try:
            for rep in range(3):
                for i, im in enumerate(imgs):
                    outs[i] = up.upscale(im)
        except Exception as e:   # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=run, args=(j,)) for j in jobs]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    for scale, up, imgs, outs in jobs:
        for i, im in enumerate(imgs):
            check(outs[i], ref.upscale(weights(scale), im), f"thread x{scale} frame {i}")
        up.close()
