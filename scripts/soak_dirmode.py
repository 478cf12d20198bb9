"""Directory mode under soak: the same 1080p directory through one context again and again for T seconds, every output file decoded
(by Pillow: an independent PNG reader) and compared with the frame's direct upscale.  Catches what a parity test on a handful of
frames cannot: a rare race between the codec pool, the lanes' rings and the pinned buffer pools.  env: T (seconds, 300), N (frames
per directory, 240), FRAMES (video | noise | toon)."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
from reve_amd import synth, ncnn_io
from reve_amd.upscaler import Upscaler, png_write
T, n = float(os.environ.get("T", "300")), int(os.environ.get("N", "240"))
kind = {"noise": synth.noise_frame, "video": synth.video_frame}.get(os.environ.get("FRAMES", "video"), synth.toon_frame)
w = synth.make_weights(2)
distinct = 24
with tempfile.TemporaryDirectory() as d, Upscaler(2, param=ncnn_io.build_param_text(2).encode(), bin=ncnn_io.build_bin(w)) as up:
    frames = [kind(i, 1920, 1080) for i in range(distinct)]
    want = [up.upscale(f) for f in frames]
    os.makedirs(d + "/in")
    for i in range(n):
        png_write(f"{d}/in/frame{i + 1:08d}.png", frames[i % distinct])
    t_end, rounds, bad, checked = time.time() + T, 0, 0, 0
    while time.time() < t_end:
        out = f"{d}/out"
        os.makedirs(out, exist_ok=True)
        order = []
        k = up.upscale_segment(d + "/in", out, on_done=lambda i, a, b: order.append(i))
        assert k == n and order == list(range(n)), (k, order[:5])
        for i in range(n):
            got = np.array(Image.open(f"{out}/frame{i + 1:08d}.png").convert("RGB"))
            checked += 1
            if not np.array_equal(got, want[i % distinct]):
                bad += 1
                print("MISMATCH round", rounds, "frame", i, int((got != want[i % distinct]).sum()), flush=True)
            os.unlink(f"{out}/frame{i + 1:08d}.png")
        rounds += 1
        if rounds % 5 == 0:
            print(f"{rounds} rounds, {checked} files checked, {bad} bad", flush=True)
print(f"SOAK_DIRMODE {'FAILED' if bad else 'OK'}: {rounds} rounds x {n} frames ({kind.__name__}), {checked} files verified by Pillow, {bad} mismatches")
