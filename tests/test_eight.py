"""GPU: eight of everything on the ONE GPU of the test box.

BASELINE's config 4 shards a stream over the 8 GPUs of a node; no such node has been available, and the largest rank / context
count exercised so far was 2.  What a one-GPU box CAN show is that eight ranks and eight in-process contexts start, bind, allocate
their rings, run and finish — under the box's CPU quota, all resolving to the same GPU's local CPUs — without deadlock, with the
right bytes, and at an aggregate rate near the single-rank one (the one GPU is the bottleneck either way).  Figures:
gpurun_out/eight_on_one_gpu.json (copied to profiles/r04/)."""
import json
import os
import time

import numpy as np
import pytest

from reve_amd import synth
from reve_amd.upscaler import Upscaler, UpscalerGroup, png_read, png_write, upscale_stream
from tests.test_gpu_parity import _bench

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _note(key, value):
    path = os.path.join(ROOT, "gpurun_out", "eight_on_one_gpu.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    d = json.load(open(path)) if os.path.exists(path) else {}
    d[key] = value
    json.dump(d, open(path, "w"), indent=1, sort_keys=True)


def test_eight_ranks_share_the_gpu():
    """`bench.py --gpus 8 --workload C4` with REVE_BENCH_BACKEND=gloo: eight processes (one per rank, as the driver launches them)
    on device 0, the control plane over gloo; every rank binds to the GPU's CPUs, allocates its pinned ring and walks its share
    of every segment.  Wall-clock budget for the whole command (eight interpreters under the box's CPU quota), and the
    aggregate pipeline rate against one rank's."""
    common = ["--steps", "40", "--warmup", "4", "--no-cpu-baseline", "--segmentsize", "80", "--min-timed-s", "2"]
    t0 = time.time()
    one = _bench(["--gpus", "1", "--workload", "C4"] + common)
    t1 = time.time()
    # (no --workload: the driver's command form; with eight ranks that is config 4's schedule)
    eight = _bench(["--gpus", "8"] + common, env={"REVE_BENCH_BACKEND": "gloo"}, timeout=900)
    t2 = time.time()
    assert eight["config"]["workload"].startswith("C4:") and [m["rank"] for m in eight["per_rank"]] == list(range(8))
    assert all(m["bound_cpus"] >= 1 and m["pipeline_fps"] > 0 and m["pinned_alloc_ms"] > 0 for m in eight["per_rank"])
    assert eight["slowest_rank"]["of_mean"] > 0.5 and eight["host_pinned_GBps"] > 0
    _note("bench_C4_1rank", one)
    _note("bench_C4_8ranks_1gpu_gloo", eight)
    _note("bench_wall_s", {"1 rank": round(t1 - t0, 1), "8 ranks": round(t2 - t1, 1)})
    assert eight["n_gpus"] == 8 and eight["config"]["frames_total"] == 8 * eight["config"]["frames_per_gpu"]
    assert eight["config"]["segments"] >= 2 and "of every segment" in eight["config"]["frame_sharding"]
    assert t2 - t1 < 300, f"eight ranks took {t2 - t1:.0f} s of wall time"
    # one GPU serves all eight processes' queues: the aggregate cannot beat one rank; how far below it falls depends on how the
    # hardware scheduler interleaves eight processes' kernels (each launch wants every CU): 0.73 and 0.91 of one rank's rate
    # were seen on two boxes.  The bound here is a floor against serialisation or starvation, not a performance claim.
    assert eight["value"] > 0.6 * one["value"], (eight["value"], one["value"])
    assert eight["pipeline_fps"] > 0.6 * one["pipeline_fps"], (eight["pipeline_fps"], one["pipeline_fps"])
    assert eight["host_placement"]["bound_cpus"] >= 1


def test_eight_contexts_in_one_process(model_bytes, tmp_path):
    """reve_create_group with the device listed eight times (the binary's `-g 0,0,0,0,0,0,0,0`): eight contexts from one parse of
    the model, eight feeder lanes each with its ring and pinned pools, all bound to the same local CPUs.  Raw frames through
    reve_upscale_stream_multi (callbacks in frame order, bytes of a single context), then a 1080p PNG directory."""
    p, b = model_bytes(2)
    with Upscaler(2, param=p, bin=b) as single, UpscalerGroup([0] * 8, 2, param=p, bin=b) as grp:
        assert len(grp.members) == 8
        frames = [synth.toon_frame(i, 640, 360) for i in range(64)]
        order = []
        t0 = time.time()
        outs = upscale_stream(grp.members, frames, on_done=order.append)
        dt = time.time() - t0
        assert order == list(range(64))
        for i in (0, 7, 8, 33, 63):
            assert np.array_equal(outs[i], single.upscale(frames[i])), i
        _note("stream_multi_8_contexts_640x360", {"frames": 64, "seconds": round(dt, 2), "note": "Python read / write callbacks"})
        assert dt < 120
        ind, outd = tmp_path / "in", tmp_path / "out"
        ind.mkdir()
        outd.mkdir()
        big = [synth.toon_frame(100 + i, 1920, 1080) for i in range(4)]
        for i in range(48):
            png_write(str(ind / f"frame{i + 1:08d}.png"), big[i % 4])
        t0 = time.time()
        n = grp.upscale_segment(str(ind), str(outd))
        dt = time.time() - t0
        assert n == 48
        for i in (0, 13, 47):
            assert np.array_equal(png_read(str(outd / f"frame{i + 1:08d}.png")), single.upscale(big[i % 4])), i
        _note("dir_multi_8_contexts_1080p_png", {"frames": 48, "seconds": round(dt, 2), "frames_per_s": round(48 / dt, 1)})
        assert dt < 120
