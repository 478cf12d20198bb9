"""Directory-mode throughput (PNG in -> PNG out through realesrgan-hip), 1080p x2, for DESIGN.md."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reve_amd import synth, ncnn_io
from reve_amd.upscaler import png_write
n = int(os.environ.get("N", "40"))
with tempfile.TemporaryDirectory() as d:
    ncnn_io.write_model(d + "/models", "realesr-animevideov3-x2", synth.make_weights(2))
    os.makedirs(d + "/in"); os.makedirs(d + "/out")
    frames = [synth.toon_frame(i, 1920, 1080) for i in range(min(n, 48))]     # (48 distinct frames, cycled: generating them is the slow part)
    for i in range(n):
        png_write(f"{d}/in/frame{i + 1:08d}.png", frames[i % len(frames)])
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "reve_amd", "realesrgan-hip")
    for tile in ("full", "0"):
        for f in os.listdir(d + "/out"):
            os.unlink(d + "/out/" + f)
        t0 = time.time()
        r = subprocess.run([exe, "-i", d + "/in", "-o", d + "/out", "-s", "2", "-m", d + "/models", "-t", tile, "-v"], capture_output=True, text=True,
                           env=dict(os.environ, REVE_DIR_STATS="1"))
        dt = time.time() - t0
        done = sum(l.endswith(" done") for l in r.stderr.splitlines())
        print("\n".join(l for l in r.stderr.splitlines() if l.startswith("[dir]")))
        print(f"-t {tile}: {done} frames in {dt:.2f} s = {done / dt:.1f} frames/s (incl. process start + model load), rc {r.returncode}", flush=True)
