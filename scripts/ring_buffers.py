"""reve_submit / reve_wait with K distinct pinned host buffers cycling (bench.py uses 3): does the three-stage overlap survive?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reve_amd import synth, ncnn_io
from reve_amd.upscaler import Upscaler, pinned_array
S, W, H = 2, 1920, 1080
w = synth.make_weights(S)
up = Upscaler(S, param=ncnn_io.build_param_text(S).encode(), bin=ncnn_io.build_bin(w))
frame = synth.toon_frame(0, W, H)
for K in (3, 8, 32):
    hin = [pinned_array((H, W, 3)) for _ in range(K)]
    hout = [pinned_array((H * S, W * S, 3)) for _ in range(K)]
    for a in hin:
        a[...] = frame
    for touch in (False, True):
        n = 300
        up.set_profiling(True); up.reset_stats()
        t0 = time.perf_counter()
        for i in range(n):
            if i >= 3:
                up.wait()
                if touch:            # a consumer reads the finished frame (what an encoder thread does)
                    hout[(i - 3) % K][::64].sum()
            up.submit(i, hin[i % K], hout[i % K])
        for _ in range(3):
            up.wait()
        dt = time.perf_counter() - t0
        st = up.stats(); k = st["ring_frames"]
        print(f"K={K:2d} touch={touch}: {n / dt:.1f} frames/s; H2D {st['h2d_ms_total']/k:.3f} chain {st['chain_ms_total']/k:.3f} D2H {st['d2h_ms_total']/k:.3f} wall {st['ring_wall_ms']/k:.3f} ms", flush=True)
