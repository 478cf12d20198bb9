"""Winograd body pairs (reve_amd/csrc/kernels_wino.hip, option "winograd") against the direct pair kernel: same library, contexts
in ONE process, interleaved rounds on one device.  First the numbers that say the kernel is right (layer probes against the
oracle's restatement of the same arithmetic, mode 4; output bytes against the direct oracle, mode 1), then per-layer device
time of the body chain and whole-frame rate.  env: N (frames per round, 30), ROUNDS (7), W, H, CHECK (1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from reve_amd import synth, ncnn_io
from reve_amd.upscaler import Upscaler
from oracle import ref

S = 2
W, H = int(os.environ.get("W", "1920")), int(os.environ.get("H", "1080"))
n = int(os.environ.get("N", "30")); rounds = int(os.environ.get("ROUNDS", "7"))
w = synth.make_weights(S)
p, b = ncnn_io.build_param_text(S).encode(), ncnn_io.build_bin(w)

if os.environ.get("CHECK", "1") == "1":
    up = Upscaler(S, param=p, bin=b)
    up.set_option("winograd", 1)
    assert up.get_option("winograd") == 1
    direct = Upscaler(S, param=p, bin=b)
    for (cw, ch) in ((48, 40), (37, 29), (64, 64), (130, 67), (200, 33), (63, 18), (125, 141)):
        img = synth.toon_frame(1, cw, ch) if (cw + ch) & 1 else synth.noise_frame(2, cw, ch)
        for layer in (2, 4, 16):
            got = up.debug_layer(img, layer)
            exp = ref.layer(w, img, layer, mode=ref.MODE_FP16_WINOGRAD_ROW)
            d1 = ref.layer(w, img, layer, mode=ref.MODE_FP16_STORAGE)
            e = np.abs(got - exp)
            bad = np.argwhere(e > 2.0 ** -6)
            print(f"{cw}x{ch} layer {layer}: vs mode 4 max |d| {e.max():.3e} mean {e.mean():.3e} (> 2^-6: {len(bad)}{' first ' + str(bad[0]) if len(bad) else ''}); "
                  f"mode 4 vs mode 1 max {np.abs(exp - d1).max():.3e}; activations max |x| {np.abs(exp).max():.2f}", flush=True)
        out = up.upscale(img).astype(int)
        o4 = ref.upscale(w, img, mode=ref.MODE_FP16_WINOGRAD_ROW).astype(int)
        o1 = ref.upscale(w, img, mode=ref.MODE_FP16_STORAGE).astype(int)
        od = direct.upscale(img).astype(int)
        print(f"{cw}x{ch} output: vs mode 4 max {np.abs(out - o4).max()} differing {float((out != o4).mean()):.3e}; vs mode 1 max {np.abs(out - o1).max()} "
              f"differing {float((out != o1).mean()):.3e}; direct kernel vs mode 1 max {np.abs(od - o1).max()} differing {float((od != o1).mean()):.3e}", flush=True)
    up.close(); direct.close()

src = torch.from_numpy(synth.noise_frame(0, W, H)).cuda()
dst = torch.empty((H * S, W * S, 3), dtype=torch.uint8, device="cuda")
ups = {}
for name, wino in (("direct pairs", 0), ("winograd pairs", 1)):
    up = Upscaler(S, param=p, bin=b)
    up.set_option("winograd", wino)
    for _ in range(3):
        up.upscale_device(src.data_ptr(), W, H, dst.data_ptr())
    up.sync()
    up.set_profiling(True)
    ups[name] = up
names = list(ups)
body = {k: [] for k in names}; fps = {k: [] for k in names}
for r in range(rounds):
    for k in (names if r % 2 == 0 else names[::-1]):
        up = ups[k]
        up.reset_stats()
        t0 = time.perf_counter()
        for _ in range(n):
            up.upscale_device(src.data_ptr(), W, H, dst.data_ptr())
        up.sync()
        dt = time.perf_counter() - t0
        st = up.stats()
        body[k].append(st["body_ms_total"] / max(st["body_launches"], 1) * 1e3)
        fps[k].append(n / dt)
for k in names:
    v = sorted(body[k]); f = sorted(fps[k])
    print(f"{W}x{H} {k:16s} body per layer median {v[len(v) // 2]:7.2f} us (min {v[0]:7.2f}, max {v[-1]:7.2f}); frames/s median {f[len(f) // 2]:7.1f} (max {f[-1]:7.1f})", flush=True)
a, c = sorted(body[names[0]]), sorted(body[names[1]])
print(f"winograd / direct per-layer time: {c[len(c) // 2] / a[len(a) // 2]:.4f}")
if H * W <= 1920 * 1080:
    img = synth.noise_frame(3, W, H)
    x = ups[names[1]].upscale(img).astype(int)
    o1 = ref.upscale(w, img).astype(int)
    print(f"{W}x{H} winograd output vs oracle mode 1: max {np.abs(x - o1).max()} LSB, differing {float((x != o1).mean()):.4e}")
