"""The `reve` CLI (reve_amd/csrc/reve_cli.cpp): reve-cli's surface + resumable segment scheduler
 above the in-process upscaler.  ffmpeg/mediainfo are absent from the image, so
tests/stubs/ stands in for them (a fake video is a JSON file; parts are .npz archives of frames)."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "reve_amd", "reve")
STUBS = os.path.join(ROOT, "tests", "stubs")


def run(args, cwd, env_extra=None, **kw):
    env = dict(os.environ, PATH=STUBS + os.pathsep + os.environ["PATH"], **(env_extra or {}))
    return subprocess.run([EXE] + args, cwd=cwd, env=env, capture_output=True, text=True, timeout=600, **kw)


def fake_video(path, frames, fps=23.976, w=48, h=32):
    with open(path, "w") as f:
        json.dump({"frames": frames, "fps": fps, "w": w, "h": h}, f)


def test_cli_validators(tmp_path):
    v = tmp_path / "in.mp4"
    fake_video(v, 5)
    t = ["--temp-dir", str(tmp_path / "t"), "--plan"]
    assert "required arguments" in run(t, tmp_path).stderr
    assert "input path not found" in run(["-i", "nope.mp4", "-s", "2", "o.mp4"] + t, tmp_path).stderr
    (tmp_path / "in.avi").write_text("x")
    assert "valid input formats: mp4/mkv" in run(["-i", "in.avi", "-s", "2", "o.mp4"] + t, tmp_path).stderr
    assert "2..=4" in run(["-i", str(v), "-s", "5", "o.mp4"] + t, tmp_path).stderr
    assert "valid output formats" in run(["-i", str(v), "-s", "2", "o.avi"] + t, tmp_path).stderr
    assert "output path already exists" in run(["-i", str(v), "-s", "2", str(v)] + t, tmp_path).stderr
    assert "0..=51" in run(["-i", str(v), "-s", "2", "-c", "52", "o.mp4"] + t, tmp_path).stderr
    assert "ultrafast/" in run(["-i", str(v), "-s", "2", "-p", "warp9", "o.mp4"] + t, tmp_path).stderr
    mkv = tmp_path / "in.mkv"
    fake_video(mkv, 5)
    assert "mkv file can only be exported as mkv" in run(["-i", str(mkv), "-s", "2", "o.mp4"] + t, tmp_path).stderr
    assert run(["--help"], tmp_path).stdout.count("--segmentsize") == 1


def test_cli_plan_segments_and_state_files(tmp_path):
    v = tmp_path / "test.mp4"
    fake_video(v, 1440, fps=23.976)     # the reference's own asset: 1440 frames, default segment size 1000
    temp = tmp_path / "temp"
    r = run(["-i", str(v), "-s", "2", "out.mp4", "--temp-dir", str(temp), "--plan"], tmp_path)
    assert r.returncode == 0, r.stderr
    video = json.loads(r.stdout.strip().splitlines()[-1])
    assert video["segments"] == [{"index": 0, "size": 1000}, {"index": 1, "size": 440}]   # no dropped frame
    assert video["frame_count"] == 1440 and video["segment_count"] == 2 and video["upscale_ratio"] == 2
    assert abs(video["frame_rate"] - 23.976) < 1e-6
    args = json.loads((temp / "args.temp").read_text())
    assert set(args) == {"inputpath", "outputpath", "scale", "segmentsize", "crf", "preset", "x265params"}
    assert args["segmentsize"] == 1000 and args["crf"] == 15 and args["preset"] == "slow"
    assert args["x265params"] == "psy-rd=2:aq-strength=1:deblock=0,0:bframes=8"
    assert os.path.isabs(args["inputpath"]) and os.path.isabs(args["outputpath"])
    assert json.loads((temp / "video.temp").read_text()) == video
    # -P (README) and -S both set the segment size; exact multiples make full segments only
    r = run(["-i", str(v), "-s", "3", "o2.mp4", "-P", "480", "--temp-dir", str(temp), "--fresh", "--plan"], tmp_path)
    video = json.loads(r.stdout.strip().splitlines()[-1])
    assert [s["size"] for s in video["segments"]] == [480, 480, 480] and video["upscale_ratio"] == 3


@pytest.mark.gpu
@pytest.mark.parametrize("io,gpu", [("png", "0"), ("pipes", "0"), ("png", "0,0"), ("pipes", "0,0")])
def test_cli_end_to_end_with_resume(tmp_path, weights, io, gpu):
    """gpu "0,0": two contexts on the one GPU of the test box = the multi-GPU code path (frames of a
    segment dealt round-robin, weights copied device-to-device)."""
    from oracle import ref
    from reve_amd import ncnn_io, synth
    models = tmp_path / "models"
    ncnn_io.write_model(str(models), "realesr-animevideov3-x2", weights(2))
    v = tmp_path / "clip.mp4"
    fake_video(v, 23, fps=24.0)
    out = tmp_path / "out.mp4"
    base = ["-i", str(v), "-s", "2", str(out), "-S", "10", "--temp-dir", str(tmp_path / "temp"), "--model-dir", str(models), "--io", io, "--gpu", gpu]
    # 1st run: the encoder "crashes" on segment 1 -> non-zero exit, state kept, segment 0's part is done
    r = run(base, tmp_path, {"REVE_STUB_FAIL_MERGE": "1"})
    assert r.returncode != 0 and not out.exists()
    state = json.loads((tmp_path / "temp" / "video.temp").read_text())
    assert [s["index"] for s in state["segments"]] == [1, 2]
    assert (tmp_path / "temp" / "video_parts" / "0.mp4").exists()
    # 2nd run resumes (reve-cli/src/main.rs:43-102): only segments 1 and 2 are redone
    r = run(["--yes", "--temp-dir", str(tmp_path / "temp"), "--model-dir", str(models), "--io", io, "--gpu", gpu], tmp_path)
    assert r.returncode == 0, r.stderr
    assert "resuming upscale" in r.stdout and "done!" in r.stdout
    assert not (tmp_path / "temp").exists()          # rebuild_temp(false) after success
    frames = np.load(out)["frames"]
    assert frames.shape == (23, 64, 96, 3)            # every source frame exactly once, in order
    for i in (0, 9, 10, 19, 20, 22):
        exp = ref.upscale(weights(2), synth.toon_frame(i, 48, 32), tile=200)   # the CLI's default = the binary's auto tiling
        d = np.abs(frames[i].astype(int) - exp.astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 0.01, i
