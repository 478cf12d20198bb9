"""Directory mode in-process, the same segment twice through one context (what the `reve` CLI does segment after segment):
the second call re-uses the pinned frame buffers the first one parked (dirmode.cpp PinnedCache)."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reve_amd import synth, ncnn_io
from reve_amd.upscaler import Upscaler, png_write
n = int(os.environ.get("N", "600"))
w = synth.make_weights(2)
with tempfile.TemporaryDirectory() as d:
    os.makedirs(d + "/in")
    # FRAMES=toon (default: flat content), video (S-video: what a decoded H.264 frame looks like — grain and block edges), noise (uniform)
    kind = {"noise": synth.noise_frame, "video": synth.video_frame}.get(os.environ.get("FRAMES", "toon"), synth.toon_frame)
    print("frames:", kind.__name__, flush=True)
    frames = [kind(i, 1920, 1080) for i in range(min(n, 48))]     # (48 distinct frames, cycled: generating them is the slow part)
    for i in range(n):
        png_write(f"{d}/in/frame{i + 1:08d}.png", frames[i % len(frames)])
    os.environ["REVE_DIR_STATS"] = "1"
    with Upscaler(2, param=ncnn_io.build_param_text(2).encode(), bin=ncnn_io.build_bin(w)) as up:
        for rnd in range(int(os.environ.get("ROUNDS", "3"))):
            out = f"{d}/out{rnd}"
            os.makedirs(out)
            up.set_profiling(True)
            up.reset_stats()
            t0 = time.time()
            k = up.upscale_segment(d + "/in", out)
            dt = time.time() - t0
            st = up.stats()
            kk = max(st["ring_frames"], 1)
            print(f"call {rnd}: {k} frames in {dt:.2f} s = {k / dt:.1f} frames/s; ring stages per frame: H2D {st['h2d_ms_total'] / kk:.3f} ms, "
                  f"chain {st['chain_ms_total'] / kk:.3f} ms, D2H {st['d2h_ms_total'] / kk:.3f} ms, ring wall {st['ring_wall_ms'] / kk:.3f} ms", flush=True)
