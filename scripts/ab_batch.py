"""Several small frames per launch (option "batch") against one frame per launch: same library, two contexts in ONE process,
interleaved rounds, frames resident on the device; then the pinned-host ring.  First the bytes: every frame of a batch against the
one-frame path.  env: SHAPES ("960x540,640x480,256x256,100x100"), N (frames per round), ROUNDS (5)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from reve_amd import synth, ncnn_io
from reve_amd.upscaler import Upscaler
S = int(os.environ.get("SCALE", "2"))
rounds = int(os.environ.get("ROUNDS", "5"))
w = synth.make_weights(S)
p, b = ncnn_io.build_param_text(S).encode(), ncnn_io.build_bin(w)
for shape in os.environ.get("SHAPES", "960x540,640x480,256x256,100x100").split(","):
    W, H = (int(x) for x in shape.split("x"))
    nf = 32
    n = int(os.environ.get("N", str(max(64, min(4096, (1920 * 1080 * 64) // (W * H) // nf * nf)))))
    src = [torch.from_numpy(synth.noise_frame(i, W, H)).cuda() for i in range(nf)]
    ups, dst = {}, {}
    for name, on in (("one per launch", 0), ("batched", 1)):
        up = Upscaler(S, param=p, bin=b)
        up.set_option("batch", on)
        dst[name] = [torch.empty((H * S, W * S, 3), dtype=torch.uint8, device="cuda") for _ in range(nf)]
        up.upscale_device_batch([t.data_ptr() for t in src], [t.data_ptr() for t in dst[name]], W, H)
        up.sync()
        up.set_profiling(True)
        ups[name] = up
    k = ups["batched"].get_option("batch_frames")
    same = all(torch.equal(a, c) for a, c in zip(dst["one per launch"], dst["batched"]))
    names = list(ups)
    fps = {x: [] for x in names}; body = {x: [] for x in names}
    for r in range(rounds):
        for x in (names if r % 2 == 0 else names[::-1]):
            up = ups[x]
            up.reset_stats()
            t0 = time.perf_counter()
            for i in range(0, n, nf):
                up.upscale_device_batch([t.data_ptr() for t in src], [t.data_ptr() for t in dst[x]], W, H)
            up.sync()
            fps[x].append(n / (time.perf_counter() - t0))
            st = up.stats()
            body[x].append(st["body_ms_total"] / max(st["body_launches"], 1) * 1e3)
    med = lambda v: sorted(v)[len(v) // 2]
    # flop roofline of the body: 2 * 36,864 MAC per pixel and layer
    frac = {x: 73728.0 * W * H * 16 * med(fps[x]) / 2.5e15 for x in names}
    print(f"{W}x{H} x{S}: batch of {k}; identical bytes: {same}; frames/s one per launch {med(fps[names[0]]):9.1f}, batched {med(fps[names[1]]):9.1f} "
          f"(x{med(fps[names[1]]) / med(fps[names[0]]):.2f}); body launch {med(body[names[0]]):7.1f} -> {med(body[names[1]]):7.1f} us; "
          f"whole-path share of the MFMA roofline {frac[names[0]]:.3f} -> {frac[names[1]]:.3f}", flush=True)
    # the pinned-host ring: submit / wait, frames collected into batches by the library
    for name in names:
        up = ups[name]
        up.set_profiling(False)
        frames = [torch.from_numpy(synth.noise_frame(i, W, H)).pin_memory().numpy() for i in range(nf)]
        outs = [torch.empty((H * S, W * S, 3), dtype=torch.uint8).pin_memory().numpy() for _ in range(nf)]
        m = min(n, 1024)
        t0 = time.perf_counter()
        inflight = 0; sub = 0; depth = 2 * k if name == "batched" else 3
        while sub < m or inflight:
            if sub < m and inflight < depth:
                up.submit(sub, frames[sub % nf], outs[sub % nf]); sub += 1; inflight += 1
            else:
                up.wait(); inflight -= 1
        dt = time.perf_counter() - t0
        ok = all(np.array_equal(outs[i], dst[name][i].cpu().numpy()) for i in range(min(nf, m)))
        print(f"    ring, {name:15s}: {m / dt:9.1f} frames/s from pinned host arrays, Python loop (ring depth {depth}); bytes equal to the device path: {ok}", flush=True)
    # a caller that keeps only THREE frames in flight whatever the batch (the `reve` CLI's lanes, reve_cli.cpp: ib[sub % 3]): reve_submit
    # holds a frame only while the GPU is busy, so such a caller never leaves the chip idle waiting for a batch it will not fill
    # (before round 5 the chain was launched only when reve_wait reached the frame)
    for name in names:
        up = Upscaler(S, param=p, bin=b, ring_depth=3)
        up.set_option("batch", 1 if name == "batched" else 0)
        m = min(n, 1024)
        t0 = time.perf_counter()
        inflight = 0; sub = 0
        while sub < m or inflight:
            if sub < m and inflight < 3:
                up.submit(sub, frames[sub % nf], outs[sub % nf]); sub += 1; inflight += 1
            else:
                up.wait(); inflight -= 1
        dt = time.perf_counter() - t0
        print(f"    ring depth 3 (the CLI's lanes), {name:15s}: {m / dt:9.1f} frames/s", flush=True)
        up.close()
    for up in ups.values():
        up.close()
