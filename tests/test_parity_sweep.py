"""Parity over WEIGHT STATISTICS.  Every other parity figure of this repository comes from one synthetic
weight draw (sigma = 0.9 / sqrt(fan_in), biases of +-0.05, slopes in [0.05, 0.3]) whose activations shrink from ~1 to ~0.1 over
the body; trained SRVGG layers have heavier tails, biases of order 1, slopes outside [0, 1] and activations of 10^2..10^3.
reve_amd.synth.WEIGHT_DRAWS holds fifteen such statistics; scripts/parity_sweep.py runs all of them x the x2 / x3 / x4 graphs
x two 512 x 288 frames on the GPU (profiles/r04/parity_sweep.txt: 90 cases).  What it found, and what is asserted here on one
frame per draw:
  * thirteen draws — activations up to 1.1e4, Student-t weights, biases of +-1, slopes in [-0.2, 1.2], 70 % zeros, shifted means:
    every output sample within 1 LSB of the oracle, 0.003 .. 4.7 % of the samples differing (the share grows with the activations'
    magnitude: an fp16 rounding flipped by the summation order is a larger step there);
  * the two EXPANDING draws (every layer's gain 1.5 / sqrt(fan_in), conv_last included: activations of 16..32 in front of an O(1)
    conv_last, 75-90 % of the output saturated) are ill-conditioned at fp16 storage for ANY evaluation order: one flipped
    rounding changes dozens of roundings in the next layer, after a few layers every activation is one ulp (2^-6) off at random,
    and conv_last sums 576 of them: +-2 LSB rms.  The HIP path differs from the oracle by up to 6-8 LSB there — and so does a
    second CPU evaluation of the oracle's own arithmetic (the torch restatement: 6 LSB, test below, needs no GPU).  The <= 1 LSB
    tolerance is a statement about well-conditioned networks; this regime is named, not hidden."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import ref
from reve_amd import ncnn_io, synth
from reve_amd.upscaler import Upscaler

EXPANDING = ("gain1.5", "student_t3_gain1.5")


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(synth.WEIGHT_DRAWS))
def test_draw_against_the_oracle(name):
    scale = 2 + sorted(synth.WEIGHT_DRAWS).index(name) % 3
    w = synth.make_weights_draw(scale, name)
    img = synth.toon_frame(7, 256, 144)
    exp = ref.upscale(w, img).astype(np.int32)
    with Upscaler(scale, param=ncnn_io.build_param_text(scale).encode(), bin=ncnn_io.build_bin(w)) as up:
        for wino in (0, 1):
            up.set_option("winograd", wino)
            d = np.abs(up.upscale(img).astype(np.int32) - exp)
            if name in EXPANDING:
                assert d.max() <= 16 and (d > 1).mean() < 0.05, (name, wino, int(d.max()))
            else:
                assert d.max() <= 1 and (d > 0).mean() < 0.08, (name, wino, int(d.max()), float((d > 0).mean()))


def test_the_expanding_regime_is_ill_conditioned_on_the_cpu_too():
    """No GPU: the oracle against the independent torch restatement of the same arithmetic (tests/golden/make_golden.py) — within
    1 LSB on a well-conditioned draw, several LSB apart on the expanding one.  What the GPU test above tolerates there is what two
    CPU evaluations already disagree by."""
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    img = synth.toon_frame(7, 96, 64)
    worst = {}
    for name in ("gain0.9_other_seed", "ramp_up_to_5e3", "everything_hot", "gain1.5"):
        w = synth.make_weights_draw(2, name)
        d = np.abs(ref.upscale(w, img).astype(np.int32) - mg.quant(mg.torch_forward(w, img, True)).astype(np.int32))
        worst[name] = int(d.max())
    assert worst["gain0.9_other_seed"] <= 1 and worst["ramp_up_to_5e3"] <= 1 and worst["everything_hot"] <= 1, worst
    assert worst["gain1.5"] >= 2, worst


def test_draws_are_pinned():
    """The draws are a pure function of their name (splitmix64 streams): a digest of each, so that a report names what it measured."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(synth.WEIGHT_DRAWS):
        h.update(synth.weights_sha256(synth.make_weights_draw(2, name)).encode())
    assert len(synth.WEIGHT_DRAWS) == 15
    assert h.hexdigest()[:16] == DRAWS_DIGEST, h.hexdigest()[:16]


DRAWS_DIGEST = "d73f101b217dc65d"
