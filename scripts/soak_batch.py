"""Soak of the several-frames-per-launch path: random small sizes, random batch fillings through reve_upscale_rgb8_device_batch
and through the submit / wait ring with random interleavings of submits and waits (partial batches released by reve_wait, the
ring running full, size changes between bursts), two contexts driven from two threads; every frame against the one-frame path.
env: ROUNDS (60), SEED (1)."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from reve_amd import synth, ncnn_io
from reve_amd.upscaler import Upscaler, ReveError

rounds = int(os.environ.get("ROUNDS", "60"))
bad = []


def worker(tid, scale):
    rng = np.random.default_rng(int(os.environ.get("SEED", "1")) * 100 + tid)
    w = synth.make_weights(scale)
    p, b = ncnn_io.build_param_text(scale).encode(), ncnn_io.build_bin(w)
    one, many = Upscaler(scale, param=p, bin=b), Upscaler(scale, param=p, bin=b)
    one.set_option("batch", 0)
    frames_done = 0
    for r in range(rounds):
        W, H = int(rng.integers(1, 700)), int(rng.integers(4, 400))
        n = int(rng.integers(1, 40))
        frames = [synth.noise_frame(r * 100 + i, W, H) if i & 1 else synth.toon_frame(r * 100 + i, W, H) for i in range(n)]
        want = [one.upscale(f) for f in frames]
        if r % 3 == 0:
            src = [torch.from_numpy(f).cuda() for f in frames]
            dst = [torch.empty((H * scale, W * scale, 3), dtype=torch.uint8, device="cuda") for _ in frames]
            many.upscale_device_batch([t.data_ptr() for t in src], [t.data_ptr() for t in dst], W, H)
            many.sync()
            got = [t.cpu().numpy() for t in dst]
        else:
            got = [np.empty((H * scale, W * scale, 3), np.uint8) for _ in frames]
            sub = ret = 0
            while ret < n:
                # submit while the dice say so and the ring takes it; otherwise retire one
                if sub < n and rng.random() < 0.7:
                    try:
                        many.submit(sub, frames[sub], got[sub])
                        sub += 1
                        continue
                    except ReveError as e:
                        if e.code != -7:
                            raise
                if sub > ret:
                    fid = many.wait()
                    if fid != ret:
                        bad.append((tid, r, "order", fid, ret))
                    ret += 1
        for i in range(n):
            if not np.array_equal(got[i], want[i]):
                bad.append((tid, r, W, H, n, i, many.get_option("batch_frames")))
        frames_done += n
    print(f"thread {tid} x{scale}: {rounds} rounds, {frames_done} frames, mismatches so far {len(bad)}", flush=True)
    one.close(); many.close()


t0 = time.time()
ths = [threading.Thread(target=worker, args=(i, s)) for i, s in enumerate((2, 4))]
for t in ths:
    t.start()
for t in ths:
    t.join()
print(bad[:10])
print("SOAK_BATCH", "FAILED" if bad else "OK", f"{time.time() - t0:.1f} s")
