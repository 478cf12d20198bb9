"""The `reve` CLI (reve_amd/csrc/reve_cli.cpp): reve-cli's surface + resumable segment scheduler
 above the in-process upscaler.  ffmpeg/mediainfo are absent from the image, so
tests/stubs/ stands in for them (a fake video is a JSON file; parts are .npz archives of frames).

Every end-to-end scenario runs in two flavours (round 5):
  gpu  the product executable on an MI355X, frames checked against the oracle (marked `gpu`);
  cpu  the SAME source (reve_cli.cpp, capi.cpp, dirmode.cpp, ... unchanged) linked over the stand-in engine of the CPU
       sanitizer builds (`make -C reve_amd/csrc san` -> build/reve_fake, AddressSanitizer + UBSan; nearest-neighbour
       "upscaling"): scheduling, state files, resume, pipes, failure paths and the tools' argv are exercised on every CPU-only
       run of the suite, with the memory checker watching the CLI's threads.  The product itself has no CPU path."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "reve_amd", "reve")
EXE_CPU = os.path.join(ROOT, "reve_amd", "csrc", "build", "reve_fake")
STUBS = os.path.join(ROOT, "tests", "stubs")
FLAVOURS = [pytest.param("gpu", marks=pytest.mark.gpu), "cpu"]
SAN_ENV = dict(ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:allocator_may_return_null=1:max_allocation_size_mb=2048",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


@pytest.fixture(scope="module")
def cpu_build():
    """build/reve_fake (make san), once per module, only when a cpu-flavoured test asks for it"""
    state = {}

    def ensure():
        if "ok" not in state:
            if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
                pytest.skip("ROCm clang++ (host sanitizer runtimes) not present")
            r = subprocess.run(["make", "-C", os.path.join(ROOT, "reve_amd", "csrc"), "san"], capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
            state["ok"] = True
    return ensure


def run(args, cwd, env_extra=None, flavour="gpu", **kw):
    env = dict(os.environ, PATH=STUBS + os.pathsep + os.environ["PATH"], **(env_extra or {}))
    if flavour == "cpu":
        env.update(SAN_ENV)
    return subprocess.run([EXE_CPU if flavour == "cpu" else EXE] + args, cwd=cwd, env=env, capture_output=True, text=True, timeout=600, **kw)


def expected(flavour, w, img, scale, **kw):
    """what the flavour's engine makes of a frame: the oracle's network, or the stand-in engine's nearest neighbour"""
    if flavour == "cpu":
        return np.repeat(np.repeat(img, scale, axis=0), scale, axis=1)
    from oracle import ref
    return ref.upscale(w, img, **kw)


def fake_video(path, frames, fps=23.976, w=48, h=32, **extra):
    with open(path, "w") as f:
        json.dump(dict({"frames": frames, "fps": fps, "w": w, "h": h}, **extra), f)


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


@pytest.mark.parametrize("flavour", FLAVOURS)
@pytest.mark.parametrize("io,gpu", [("png", "0"), ("pipes", "0"), ("png", "0,0"), ("pipes", "0,0")])
def test_cli_end_to_end_with_resume(tmp_path, weights, io, gpu, flavour, cpu_build):
    """gpu "0,0": two contexts on the one GPU of the test box = the multi-GPU code path (frames of a
    segment dealt round-robin, weights copied device-to-device)."""
    from reve_amd import ncnn_io, synth
    if flavour == "cpu":
        cpu_build()
    models = tmp_path / "models"
    ncnn_io.write_model(str(models), "realesr-animevideov3-x2", weights(2))
    v = tmp_path / "clip.mp4"
    fake_video(v, 23, fps=24.0)
    out = tmp_path / "out.mp4"
    base = ["-i", str(v), "-s", "2", str(out), "-S", "10", "--temp-dir", str(tmp_path / "temp"), "--model-dir", str(models), "--io", io, "--gpu", gpu]
    # 1st run: the encoder "crashes" on segment 1 -> non-zero exit, state kept, segment 0's part is done
    r = run(base, tmp_path, {"REVE_STUB_FAIL_MERGE": "1"}, flavour)
    assert r.returncode != 0 and not out.exists()
    state = json.loads((tmp_path / "temp" / "video.temp").read_text())
    left = [s["index"] for s in state["segments"]]
    # (pipes over several GPUs run whole segments in parallel lanes: the lane that has finished segment 0 may have taken and finished
    # segment 2 before segment 1's encoder died — more done, and just as resumable)
    assert left == [1, 2] or (io == "pipes" and "," in gpu and left == [1]), left
    assert (tmp_path / "temp" / "video_parts" / "0.mp4").exists()
    # 2nd run resumes (reve-cli/src/main.rs:43-102): only segments 1 and 2 are redone
    r = run(["--yes", "--temp-dir", str(tmp_path / "temp"), "--model-dir", str(models), "--io", io, "--gpu", gpu], tmp_path, None, flavour)
    assert r.returncode == 0, r.stderr
    assert "resuming upscale" in r.stdout and "done!" in r.stdout
    assert not (tmp_path / "temp").exists()          # rebuild_temp(false) after success: reve's own files gone, dir empty -> removed
    frames = np.load(out)["frames"]
    assert frames.shape == (23, 64, 96, 3)            # every source frame exactly once, in order
    for i in (0, 9, 10, 19, 20, 22):
        exp = expected(flavour, weights(2), synth.toon_frame(i, 48, 32), 2, tile=200)   # the CLI's default = the binary's auto tiling
        d = np.abs(frames[i].astype(int) - exp.astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 0.01, i


@pytest.mark.parametrize("flavour", FLAVOURS)
@pytest.mark.parametrize("io", ["png", "pipes"])
@pytest.mark.parametrize("exact_rate", [False, True])
def test_cli_ntsc_rate_loses_no_frame(tmp_path, weights, io, exact_rate, flavour, cpu_build):
    """A 24000/1001 clip whose rate mediainfo prints as '23.976': seeking to exactly N/23.976 lands microseconds AFTER
    frame N and an accurate-seek ffmpeg drops it (every later segment shifts, the last comes up short).  The stub models
    that seek; every source frame must arrive exactly once, with and without an exact rational rate from the container."""
    from reve_amd import ncnn_io, synth
    if flavour == "cpu":
        cpu_build()
    models = tmp_path / "models"
    ncnn_io.write_model(str(models), "realesr-animevideov3-x2", weights(2))
    v = tmp_path / "ntsc.mp4"
    # fps_num / fps_den: the container's true rate (what the ffmpeg stub's timestamps follow); mediainfo prints
    # "23.976" and reports the rational only when asked to
    fake_video(v, 450, fps=23.976, w=16, h=12, fps_num=24000, fps_den=1001, mediainfo_reports_rational=exact_rate)
    out = tmp_path / "out.mp4"
    r = run(["-i", str(v), "-s", "2", str(out), "-S", "200", "--temp-dir", str(tmp_path / "temp"), "--model-dir", str(models), "--io", io, "--tile", "full"], tmp_path, None, flavour)
    assert r.returncode == 0, r.stderr[-2000:]
    frames = np.load(out)["frames"]
    assert frames.shape == (450, 24, 32, 3)
    for i in (0, 199, 200, 201, 399, 400, 449):      # around both segment boundaries
        exp = expected(flavour, weights(2), synth.toon_frame(i, 16, 12), 2)
        assert np.abs(frames[i].astype(int) - exp.astype(int)).max() <= 1, i


@pytest.mark.parametrize("flavour", FLAVOURS)
@pytest.mark.parametrize("io", ["png", "pipes"])
def test_cli_short_last_segment_is_tolerated(tmp_path, weights, io, flavour, cpu_build):
    """mediainfo's FrameCount can exceed what the container really holds by a frame: a LAST segment that ends early is
    accepted (with a note), the frames that exist all arrive."""
    from reve_amd import ncnn_io
    if flavour == "cpu":
        cpu_build()
    models = tmp_path / "models"
    ncnn_io.write_model(str(models), "realesr-animevideov3-x2", weights(2))
    v = tmp_path / "clip.mp4"
    fake_video(v, 25, fps=24.0, w=16, h=12, actual_frames=24)
    out = tmp_path / "out.mp4"
    r = run(["-i", str(v), "-s", "2", str(out), "-S", "10", "--temp-dir", str(tmp_path / "temp"), "--model-dir", str(models), "--io", io], tmp_path, None, flavour)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "before its declared length" in r.stderr
    assert np.load(out)["frames"].shape == (24, 24, 32, 3)


@pytest.mark.parametrize("flavour", FLAVOURS)
def test_cli_decoder_death_in_the_last_segment_is_a_failure(tmp_path, weights, flavour, cpu_build):
    """Pipe transport: a short read on the last segment is only 'the container holds a frame less than declared' when the decoder
    exited cleanly.  A decoder that dies in mid-stream must leave a failed run with its state kept — not a truncated part that is
    checkpointed, concatenated and reported as success."""
    from reve_amd import ncnn_io
    if flavour == "cpu":
        cpu_build()
    models = tmp_path / "models"
    ncnn_io.write_model(str(models), "realesr-animevideov3-x2", weights(2))
    v = tmp_path / "clip.mp4"
    fake_video(v, 25, fps=24.0, w=16, h=12)
    out = tmp_path / "out.mp4"
    base = ["-i", str(v), "-s", "2", str(out), "-S", "10", "--temp-dir", str(tmp_path / "temp"), "--model-dir", str(models), "--io", "pipes"]
    r = run(base, tmp_path, {"REVE_STUB_DECODER_DIES_AT": "23"}, flavour)     # frame 3 of the 5-frame last segment
    assert r.returncode != 0 and "decoder died" in r.stderr, r.stderr[-2000:]
    assert not out.exists()
    state = json.loads((tmp_path / "temp" / "video.temp").read_text())
    assert 2 in [s["index"] for s in state["segments"]]                 # the damaged segment is still to do (earlier ones may or may not have been reaped yet)
    r = run(["--yes", "--temp-dir", str(tmp_path / "temp"), "--model-dir", str(models), "--io", "pipes"], tmp_path, None, flavour)   # resume: the whole clip arrives
    assert r.returncode == 0, r.stderr[-2000:]
    assert np.load(out)["frames"].shape == (25, 24, 32, 3)
    # a clean exit that far short of the declared length is not "a frame now and then" either
    v2 = tmp_path / "clip2.mp4"
    fake_video(v2, 25, fps=24.0, w=16, h=12, actual_frames=15)
    r = run(["-i", str(v2), "-s", "2", str(tmp_path / "o2.mp4"), "-S", "10", "--temp-dir", str(tmp_path / "temp2"), "--model-dir", str(models), "--io", "pipes"], tmp_path, None, flavour)
    assert r.returncode != 0


def test_cli_never_wipes_a_users_directory(tmp_path):
    """--temp-dir may be any directory: a fresh run removes only what reve itself creates there."""
    v = tmp_path / "in.mp4"
    fake_video(v, 30)
    temp = tmp_path / "work"
    (temp / "video_parts").mkdir(parents=True)
    (temp / "video_parts" / "0.mp4").write_text("stale part")
    (temp / "args.temp").write_text("{}")
    (temp / "thesis.tex").write_text("precious")
    (temp / "photos").mkdir()
    (temp / "photos" / "a.jpg").write_text("precious")
    r = run(["-i", str(v), "-s", "2", "o.mp4", "--temp-dir", str(temp), "--fresh", "--plan"], tmp_path)
    assert r.returncode == 0, r.stderr
    assert (temp / "thesis.tex").read_text() == "precious" and (temp / "photos" / "a.jpg").read_text() == "precious"
    assert not (temp / "video_parts" / "0.mp4").exists()          # reve's own stale state is gone
    assert json.loads((temp / "args.temp").read_text())["segmentsize"] == 1000


@pytest.mark.parametrize("flavour", FLAVOURS)
@pytest.mark.parametrize("io", ["png", "pipes"])
def test_cli_failures_leave_from_the_main_thread(tmp_path, weights, io, flavour, cpu_build):
    """A tool that fails on a worker thread (export of segment 1, merge of segment 0) must not take the process down
    from that thread: exit status 1, 'state kept', state files still parseable, no crash signal."""
    from reve_amd import ncnn_io
    if flavour == "cpu":
        cpu_build()
    models = tmp_path / "models"
    ncnn_io.write_model(str(models), "realesr-animevideov3-x2", weights(2))
    v = tmp_path / "clip.mp4"
    fake_video(v, 30, fps=24.0, w=16, h=12)
    for env in ({"REVE_STUB_FAIL_MERGE": "0"}, {"REVE_STUB_FAIL_EXPORT": "1"}):
        temp = tmp_path / ("temp_" + "_".join(env))
        r = run(["-i", str(v), "-s", "2", str(tmp_path / "o.mp4"), "-S", "10", "--temp-dir", str(temp), "--model-dir", str(models), "--io", io], tmp_path, env, flavour)
        assert r.returncode == 1, (r.returncode, r.stderr[-1500:])
        assert "state kept" in r.stderr and "error:" in r.stderr
        state = json.loads((temp / "video.temp").read_text())
        assert state["segments"] and state["segments"][0]["index"] in (0, 1)
        assert not (tmp_path / "o.mp4").exists()


@pytest.mark.parametrize("flavour", FLAVOURS)
def test_cli_tool_command_lines_match_the_reference(tmp_path, weights, flavour, cpu_build):
    """The exact ffmpeg argument lists of reve-shared/src/lib.rs:100-119 (export), reve-cli/src/main.rs:306-326 (merge) and
    lib.rs:181-204 (concat), with portable paths; -ss carries half a frame of slack instead of the reference's whole frame
    and uses the container's exact rate when mediainfo reports one."""
    from reve_amd import ncnn_io
    if flavour == "cpu":
        cpu_build()
    models = tmp_path / "models"
    ncnn_io.write_model(str(models), "realesr-animevideov3-x3", weights(3))
    v = tmp_path / "clip.mkv"
    fake_video(v, 25, fps=23.976, w=16, h=12, fps_num=24000, fps_den=1001, mediainfo_reports_rational=True)
    out = tmp_path / "out.mkv"
    temp = tmp_path / "temp"
    logf = tmp_path / "argv.log"
    r = run(["-i", str(v), "-s", "3", str(out), "-S", "10", "-c", "18", "-p", "fast", "--temp-dir", str(temp), "--model-dir", str(models)],
            tmp_path, {"REVE_STUB_ARGV_LOG": str(logf)}, flavour)
    assert r.returncode == 0, r.stderr[-2000:]
    calls = [json.loads(l) for l in logf.read_text().splitlines()]
    exports = [c for c in calls if "-vframes" in c]
    merges = [c for c in calls if "image2" in c]
    concat = [c for c in calls if "concat" in c]
    assert len(exports) == 3 and len(merges) == 3 and len(concat) == 1
    t = str(temp)
    for i, (c, n) in enumerate(zip(sorted(exports, key=lambda c: float(c[3])), (10, 10, 5))):
        ss = "0" if i == 0 else "%.6f" % ((i * 10 - 0.5) * 1001 / 24000)
        assert c == ["-v", "verbose", "-ss", ss, "-i", str(v), "-qscale:v", "1", "-qmin", "1", "-qmax", "1", "-vsync", "0",
                     "-vframes", str(n), f"{t}/tmp_frames/{i}/frame%08d.png"]                    # lib.rs:100-119
    for i, c in enumerate(sorted(merges, key=lambda c: c[-1])):
        assert c == ["-v", "verbose", "-f", "image2", "-framerate", "23.976/1", "-i", f"{t}/out_frames/{i}/frame%08d.png",
                     "-c:v", "libx265", "-pix_fmt", "yuv420p10le", "-crf", "18", "-preset", "fast",
                     "-x265-params", "psy-rd=2:aq-strength=1:deblock=0,0:bframes=8", f"{t}/video_parts/{i}.mp4"]   # main.rs:306-326
    assert concat[0] == ["-f", "concat", "-safe", "0", "-i", f"{t}/parts.txt", "-i", str(v), "-map", "0:v", "-map", "1:a?",
                         "-map", "1:s?", "-map_chapters", "1", "-c", "copy", str(out)]           # lib.rs:181-204
    assert np.load(out)["frames"].shape == (25, 36, 48, 3)
