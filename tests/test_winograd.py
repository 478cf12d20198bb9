"""GPU: the Winograd body-pair kernel (reve_amd/csrc/kernels_wino.hip, `reve_set_option("winograd", 1)`; what the default, auto, chooses
for well-conditioned weights since round 6).

It evaluates the 16 body layers by F(2,3) along the row nested with the direct sum over the tap rows: two thirds of the MFMAs, a
different (and differently rounded) sum — so it is NOT bit-identical to the direct kernels.  It is held to two bars:
  * against the oracle's restatement of exactly its arithmetic (mode 4, oracle/srvgg_ref.c): activations agree to the fp16 grid
    (the only freedom left is the MFMA's internal summation order);
  * against the direct oracle (mode 1, the parity target of the whole path): every 8-bit output sample within 1 LSB, under 1 %
    of them differing — the same tolerance as the direct kernels.
0.90 of the direct pair kernel's time at 1080p (+8-10 % frames/s on noise frames, +15.9 % on flat ones: profiles/r04/ab_wino.txt,
bench_toon_vs_noise.txt, profiles/r05/bench_box_spread.txt)."""
import numpy as np
import pytest

from oracle import ref
from reve_amd import synth
from reve_amd.upscaler import Upscaler

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wino(model_bytes):
    ups = {}

    def get(scale, on=1, tile=0):
        if (scale, on, tile) not in ups:
            p, b = model_bytes(scale)
            up = Upscaler(scale, param=p, bin=b, tile=tile)
            up.set_option("winograd", on)
            assert up.get_option("winograd") == on
            ups[(scale, on, tile)] = up
        return ups[(scale, on, tile)]

    yield get
    for up in ups.values():
        up.close()


def test_auto_by_default_and_the_environment_pins_it(model_bytes, monkeypatch):
    """Round 6: the library's default is auto (mode 2) and the synthetic weights pass the rule; REVE_WINOGRAD=0 in the environment —
    what a deployment whose caller cannot change sets (reve passes no options, reve-shared/src/lib.rs:134-147) — pins the direct
    kernels at reve_create, =1 forces Winograd, an empty value changes nothing."""
    p, b = model_bytes(2)
    monkeypatch.delenv("REVE_WINOGRAD", raising=False)
    img = synth.noise_frame(3, 200, 120)
    with Upscaler(2, param=p, bin=b) as up:
        assert up.get_option("winograd_mode") == 2 and up.get_option("winograd") == 1 and up.get_option("winograd_kappa_permille") < 500
        auto = up.upscale(img)
        up.set_option("winograd", 0)
        direct = up.upscale(img)
        assert not np.array_equal(auto, direct) and np.abs(auto.astype(int) - direct).max() <= 1
    for env, mode, on in (("0", 0, 0), ("1", 1, 1), ("auto", 2, 1), ("", 2, 1), ("off", 0, 0), ("on", 1, 1), ("direct", 0, 0), ("banana", 2, 1)):
        monkeypatch.setenv("REVE_WINOGRAD", env)
        with Upscaler(2, param=p, bin=b) as up:
            assert (up.get_option("winograd_mode"), up.get_option("winograd")) == (mode, on), env
            assert np.array_equal(up.upscale(img), auto if on else direct)


def test_auto_mode_follows_the_weights_conditioning(model_bytes):
    """`reve_set_option("winograd", 2)`: the Winograd pairs run if and only if the loaded weights' conditioning estimate
    (reve_amd/csrc/model.h: the fp16 storage noise they carry to the 8-bit output, in LSB rms) is under 0.5.  The thirteen
    well-conditioned draws of the parity sweep choose Winograd and stay within 1 LSB of the oracle; the two expanding draws —
    where no two evaluation orders agree to 1 LSB, the CPU restatements included — choose the direct kernels."""
    from reve_amd import ncnn_io
    from tests.test_parity_sweep import EXPANDING
    img = synth.toon_frame(7, 256, 144)
    chose = {}
    for name in sorted(synth.WEIGHT_DRAWS):
        scale = 2 + sorted(synth.WEIGHT_DRAWS).index(name) % 3
        w = synth.make_weights_draw(scale, name)
        with Upscaler(scale, param=ncnn_io.build_param_text(scale).encode(), bin=ncnn_io.build_bin(w)) as up:
            assert up.get_option("winograd_mode") == 2          # (the default)
            up.set_option("winograd", 0)
            direct = up.upscale(img)
            up.set_option("winograd", 2)
            assert up.get_option("winograd_mode") == 2
            chose[name] = up.get_option("winograd")
            kappa = up.get_option("winograd_kappa_permille")
            assert (kappa < 500) == bool(chose[name]), (name, kappa)
            out = up.upscale(img)
            if name in EXPANDING:
                assert chose[name] == 0 and np.array_equal(out, direct), name          # auto fell back to the direct kernels: same bytes
            else:
                d = np.abs(out.astype(np.int32) - ref.upscale(w, img).astype(np.int32))
                assert chose[name] == 1 and d.max() <= 1 and (d > 0).mean() < 0.08, (name, int(d.max()))
                assert not np.array_equal(out, direct) or name == "ramp_up_then_down"     # (it really is the other evaluation)
            up.set_option("winograd", 0)
            assert up.get_option("winograd") == 0 and np.array_equal(up.upscale(img), direct)
    assert sum(chose.values()) == 13, chose


# around the strip width (62 valid columns), odd widths (the last tile's second pixel lies outside the frame), odd heights,
# one-row segments, a frame smaller than a tile, many strips x segments, more strips than CUs (a second unit per workgroup)
SHAPES = [(48, 40), (37, 29), (1, 1), (3, 2), (62, 9), (63, 40), (125, 21), (130, 67), (200, 131), (640, 360), (16100, 20)]


@pytest.mark.parametrize("w,h", SHAPES)
def test_activations_against_the_restated_arithmetic(wino, weights, w, h):
    img = synth.noise_frame(w * 1000 + h, w, h)
    up = wino(2)
    for layer in (2, 16):
        got = up.debug_layer(img, layer)
        exp = ref.layer(weights(2), img, layer, mode=ref.MODE_FP16_WINOGRAD_ROW)
        d = np.abs(got - exp)
        # one step of the fp16 grid at the activations' magnitude (values below 2 here), mean far below it
        assert d.max() <= 2.0 ** -9 * max(1.0, float(np.abs(exp).max())), (w, h, layer, float(d.max()), np.argwhere(d > 2.0 ** -9)[:5].tolist())
        assert d.mean() < 2.0 ** -13, (w, h, layer, float(d.mean()))


@pytest.mark.parametrize("scale", [2, 3, 4])
def test_frames_within_one_lsb_of_the_direct_oracle(wino, weights, scale):
    for w, h in ((150, 97), (64, 64), (333, 120)):
        img = synth.toon_frame(scale * 7 + w, w, h) if w != 64 else synth.noise_frame(scale, w, h)
        out = wino(scale).upscale(img).astype(np.int32)
        d1 = np.abs(out - ref.upscale(weights(scale), img).astype(np.int32))
        d4 = np.abs(out - ref.upscale(weights(scale), img, mode=ref.MODE_FP16_WINOGRAD_ROW).astype(np.int32))
        assert d1.max() <= 1 and (d1 > 0).mean() < 0.01, (scale, w, h, int(d1.max()), float((d1 > 0).mean()))
        assert d4.max() <= 1 and (d4 > 0).mean() < 0.01, (scale, w, h, int(d4.max()), float((d4 > 0).mean()))


def test_ring_graph_and_repeatability(wino, weights):
    """The same bytes through the synchronous call, the submit / wait ring and the ring replayed as a captured graph."""
    up = wino(2)
    frames = [synth.toon_frame(i, 192, 108) for i in range(5)]
    want = [up.upscale(f) for f in frames]
    for graph in (0, 1):
        up.set_option("graph", graph)
        outs = [np.empty((216, 384, 3), np.uint8) for _ in frames]
        done = []
        for i, f in enumerate(frames):
            if i >= 3:
                done.append(up.wait())
            up.submit(i, f, outs[i])
        while len(done) < len(frames):
            done.append(up.wait())
        assert done == list(range(len(frames)))
        for a, b in zip(outs, want):
            assert np.array_equal(a, b)
    up.set_option("graph", 0)


def test_tiled_and_stacked_frames(wino, weights, model_bytes):
    """Several planes on one canvas — the binary's tiling, small frames that share their launches — go through the Winograd
    kernel's canvas instantiation (gutter columns and rows stay zero): within 1 LSB of the direct oracle WITH the same tiling, and
    a batch of frames gives the bytes of the same frames one by one."""
    for (w, h, tile) in ((300, 170, 100), (640, 360, 200), (130, 500, 64)):
        img = synth.toon_frame(4, w, h)
        out = wino(2, 1, tile).upscale(img).astype(np.int32)
        d = np.abs(out - ref.upscale(weights(2), img, tile=tile).astype(np.int32))
        assert d.max() <= 1 and (d > 0).mean() < 0.01, (w, h, tile, int(d.max()), float((d > 0).mean()))
        direct = wino(2, 0, tile).upscale(img).astype(np.int32)
        assert np.abs(out - direct).max() <= 1
    import torch
    up = wino(2)
    for (w, h, n) in ((100, 100, 16), (256, 144, 5), (333, 120, 3)):
        frames = [synth.noise_frame(100 + i, w, h) if i & 1 else synth.toon_frame(100 + i, w, h) for i in range(n)]
        one = [up.upscale(f) for f in frames]
        src = [torch.from_numpy(f).cuda() for f in frames]
        dst = [torch.empty((2 * h, 2 * w, 3), dtype=torch.uint8, device="cuda") for _ in frames]
        up.upscale_device_batch([s.data_ptr() for s in src], [d.data_ptr() for d in dst], w, h)
        up.sync()
        for a, b, f in zip(dst, one, frames):
            assert np.array_equal(a.cpu().numpy(), b), (w, h, n)
            d = np.abs(b.astype(np.int32) - ref.upscale(weights(2), f).astype(np.int32))
            assert d.max() <= 1


def test_1080p_frame_against_both_oracles(wino, weights):
    img = synth.noise_frame(11, 1920, 1080)
    out = wino(2).upscale(img).astype(np.int32)
    d1 = np.abs(out - ref.upscale(weights(2), img).astype(np.int32))
    assert d1.max() <= 1 and (d1 > 0).mean() < 0.01, (int(d1.max()), float((d1 > 0).mean()))
