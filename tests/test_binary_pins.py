"""CPU: the ingestion path for reference-held vectors, REHEARSED.

Nothing in /root/reference pins the arithmetic (reve-cli/tests/run_test.rs:31-34 checks that out.mp4 exists) and the original
binary with its model files cannot run here, so `parity` stays "unpinned" until a maintainer runs the binary once.  This test
makes sure that day is one command: scripts/pin_against_binary.py over a directory of inputs and a directory of "the binary's"
outputs — here the outputs are written by the oracle's mode 3 (fp16 storage, Winograd F(4x4,3x3): a plausible guess at what ncnn's
Vulkan path does, NOT the binary; the pin says so) with 32-pixel tiles — must find the tiling, rank the evaluations, write a pin,
and the suites must consume the pin when REVE_MODEL_DIR names the same model files, and skip it when it does not.
"""
import json
import os
import subprocess
import sys

import numpy as np

from oracle import ref
from reve_amd import ncnn_io, synth
from reve_amd.upscaler import png_write

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pin_against_binary_rehearsal(tmp_path):
    models, ind, outd, pins = tmp_path / "models", tmp_path / "in", tmp_path / "out", tmp_path / "pins"
    for d in (models, ind, outd):
        d.mkdir()
    w = synth.make_weights(2, seed=0x51A0D)
    ncnn_io.write_model(str(models), "realesr-animevideov3-x2", w, fp16=True)
    wq = ncnn_io.parse_model(*(lambda pb: (pb[0].decode(), pb[1]))(ncnn_io.read_model_files(str(models), "realesr-animevideov3-x2")))
    frames = [synth.toon_frame(3, 90, 70), synth.noise_frame(4, 90, 70)]
    for i, f in enumerate(frames):
        png_write(str(ind / f"frame{i + 1:08d}.png"), f)
        png_write(str(outd / f"frame{i + 1:08d}.png"), ref.upscale(wq, f, mode=ref.MODE_FP16_WINOGRAD43, tile=32, prepad=10))
    png_write(str(ind / "frame00000009.png"), frames[0])          # (no counterpart in the output directory: ignored)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pin_against_binary.py"), "--model-dir", str(models), "--frames", str(ind),
                        "--binary-out", str(outd), "--tiles", "0,32,64", "--no-gpu", "--pins-dir", str(pins), "--name", "rehearsal",
                        "--stand-in", "oracle mode 3 standing in for the binary (tests/test_binary_pins.py)"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rep = json.load(open(pins / "rehearsal.report.json"))
    assert rep["tile_consistent_with_the_binary"] == 32 and rep["tile_search"]["32"]["differing_fraction"] < rep["tile_search"]["0"]["differing_fraction"]
    s = rep["summary"]
    assert s["oracle_mode3_fp16_winograd_f4x4"] == {"max_lsb": 0, "worst_differing_fraction": 0.0, "within_tolerance": True}
    assert rep["closest_evaluation"] == "oracle_mode3_fp16_winograd_f4x4" and rep["verdict"].startswith("PINNED")
    assert s["oracle_mode1_fp16_direct"]["max_lsb"] <= 1 and 0 < s["oracle_mode1_fp16_direct"]["worst_differing_fraction"] < 0.02
    assert 0.001 < rep["kappa"] < 0.5 and rep["evaluation_auto_would_choose"].startswith("winograd") and "standing in" in rep["binary_out_is"]
    assert rep["frames"] == ["frame00000001.png", "frame00000002.png"] and "not run" in rep["hip"]
    z = np.load(pins / "rehearsal.npz")
    meta = json.loads(str(z["meta"]))
    assert meta["tile"] == 32 and set(meta["model_sha256"]) == {"realesr-animevideov3-x2.param", "realesr-animevideov3-x2.bin"}
    assert np.array_equal(z["img_1"], frames[1]) and z["out_0"].shape == (140, 180, 3)
    assert set(z.files) == {"meta", "img_0", "out_0", "img_1", "out_1"}          # data only: no model bytes in the pin

    # the suites consume the pin when REVE_MODEL_DIR holds the model it names ...
    env = dict(os.environ, REVE_BINARY_PINS=str(pins), REVE_MODEL_DIR=str(models))
    t = subprocess.run([sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_oracle.py"), "-k", "test_binary_pins"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert t.returncode == 0 and "1 passed" in t.stdout, t.stdout[-2000:]
    # ... skip it when the model at hand is another one (a different seed here), and when no model directory is given
    other = tmp_path / "other"
    other.mkdir()
    ncnn_io.write_model(str(other), "realesr-animevideov3-x2", synth.make_weights(2), fp16=True)
    for e in (dict(env, REVE_MODEL_DIR=str(other)), {k: v for k, v in env.items() if k != "REVE_MODEL_DIR"}):
        t = subprocess.run([sys.executable, "-m", "pytest", "-q", "-rs", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_oracle.py"), "-k", "test_binary_pins"],
                           capture_output=True, text=True, timeout=600, cwd=ROOT, env=e)
        assert t.returncode == 0 and "1 skipped" in t.stdout and "parity unpinned" in t.stdout, t.stdout[-2000:]
    # a tool given outputs that are NOT within tolerance says so with status 1 (here: outputs of another model)
    bad = tmp_path / "bad"
    bad.mkdir()
    w_other = synth.make_weights(2)
    for i, f in enumerate(frames):
        png_write(str(bad / f"frame{i + 1:08d}.png"), ref.upscale(w_other, f))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pin_against_binary.py"), "--model-dir", str(models), "--frames", str(ind),
                        "--binary-out", str(bad), "--tile", "0", "--no-gpu", "--pins-dir", str(tmp_path / "pins2")], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 1 and "NOT within 1 LSB" in r.stdout
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pin_against_binary.py"), "--model-dir", str(tmp_path), "--frames", str(ind),
                        "--binary-out", str(outd), "--no-gpu"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 2 and "missing model file" in r.stderr
    # --binary EXE: the tool starts the executable itself with reve's argv (lib.rs:134-147) and reads what it wrote.  The stand-in here
    # checks the argv it is given and copies the outputs made above (the real binary needs a Vulkan GPU; the CPU stand-in engine of
    # the sanitizer builds upscales by nearest neighbour and would only prove "not within tolerance").
    stub = tmp_path / "realesrgan-ncnn-vulkan"
    stub.write_text("#!/bin/sh\n"
                    f"[ \"$1\" = -i ] && [ \"$3\" = -o ] && [ \"$5 $6 $7 $8 $9 ${{10}}\" = \"-n realesr-animevideov3-x2 -s 2 -f png\" ] && [ \"${{11}}\" = -m ] || exit 9\n"
                    f"cp {outd}/*.png \"$4\"/\n")
    stub.chmod(0o755)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pin_against_binary.py"), "--model-dir", str(models), "--frames", str(ind),
                        "--binary", str(stub), "--tiles", "32", "--no-gpu", "--pins-dir", str(tmp_path / "pins3"), "--name", "ran_it"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and os.path.exists(tmp_path / "pins3" / "ran_it.npz"), r.stdout[-1500:] + r.stderr[-1500:]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pin_against_binary.py"), "--model-dir", str(models), "--frames", str(ind), "--no-gpu"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 2 and "exactly one of" in r.stderr
