"""CPU: frame sharding across ranks incl. a world_size-2 gloo run.

The path has no data-path collective: frames of a segment are independent, rank r of G takes
frames r, r+G, ...; the only exchange is the one-off broadcast of the model bytes from rank 0.
On CPU the per-rank upscaler is stood in for by the oracle (tests may use it; the product never does).
"""
import os
import socket
import sys

import numpy as np
import pytest

from reve_amd import shard, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_frames_for_rank_partition():
    for n in (0, 1, 7, 8, 1000, 1001):
        for g in (1, 2, 3, 4, 8):
            parts = [shard.frames_for_rank(n, r, g) for r in range(g)]
            flat = sorted(i for p in parts for i in p)
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    with pytest.raises(ValueError):
        shard.frames_for_rank(10, 2, 2)


def test_segment_shards_keep_segment_granularity():
    # 8000-frame stream, segmentsize 1000 (BASELINE config 4): every segment is spread over all ranks
    segs = shard.segments(8000, 1000)
    assert len(segs) == 8 and all(s.size == 1000 for s in segs)
    segs = shard.segments(1440, 1000)   # the reference's test.mp4: 1440 frames -> 1000 + 440
    assert [(s.index, s.start, s.size) for s in segs] == [(0, 0, 1000), (1, 1000, 440)]
    assert sum(s.size for s in shard.segments(1001, 1000)) == 1001   # no dropped frame
    assert shard.segments(0, 1000) == []


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import ref
    from reve_amd import ncnn_io
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # rank 0 owns the model; the others receive its bytes (RCCL broadcast on GPUs, gloo here)
        if rank == 0:
            w = synth.make_weights(2)
            blobs = [ncnn_io.build_param_text(2).encode(), ncnn_io.build_bin(w)]
        else:
            blobs = [None, None]
        param, binb = shard.broadcast_model(blobs[0], blobs[1], src=0)
        w = ncnn_io.parse_model(param.decode(), binb)
        n_frames = 5
        mine = shard.frames_for_rank(n_frames, rank, world)
        outs = {i: ref.upscale(w, synth.noise_frame(i, 24, 16), nthreads=1) for i in mine}
        gathered = shard.gather_results(outs, n_frames, dst=0)
        total = shard.all_reduce_sum(float(len(mine)))
        q.put((rank, len(mine), total, None if gathered is None else [g.tobytes() for g in gathered]))
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [3, 2] and all(r[2] == 5.0 for r in res)
    from oracle import ref
    w = synth.make_weights(2)
    expect = [ref.upscale(w, synth.noise_frame(i, 24, 16), nthreads=1).tobytes() for i in range(5)]
    assert res[0][3] == expect and res[1][3] is None


def test_bench_refuses_to_mislabel_a_run():
    """bench.py without a GPU: no line, non-zero exit (no CPU fallback); `--gpus N` outside torchrun must not quietly run one
    rank, and a WORLD_SIZE that disagrees with --gpus is an error before anything is measured."""
    import subprocess
    if __import__("torch").cuda.device_count() > 0:
        pytest.skip("covered on the GPU by tests/test_gpu_parity.py::test_bench_launches_its_own_ranks")
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    for args, extra in ((["--gpus", "1"], {}), (["--gpus", "2"], {}), (["--gpus", "2"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}),
                        (["--gpus", "0"], {})):
        r = subprocess.run([sys.executable, bench] + args, capture_output=True, text=True, timeout=300, cwd=ROOT, env=dict(env, **extra))
        assert r.returncode != 0 and "{" not in r.stdout, (args, extra, r.stdout[-300:], r.stderr[-300:])


def test_c4_segment_schedule_covers_every_frame_once():
    """bench.py --workload C4: rank r's share of each 1000-frame segment (frames r, r+G, ...) — all ranks together cover the
    8000-frame stream exactly once, segment by segment."""
    for world in (1, 2, 4, 8):
        segs = shard.segments(8000, 1000)
        seen = []
        for sg in segs:
            per_rank = [shard.frames_for_rank(sg.size, r, world) for r in range(world)]
            assert sum(len(p) for p in per_rank) == sg.size and max(len(p) for p in per_rank) - min(len(p) for p in per_rank) <= 1
            seen += sorted(sg.start + f for p in per_rank for f in p)
        assert seen == list(range(8000))


def test_control_plane_is_all_that_differs_between_rccl_and_gloo():
    """bench.py's rank path over nccl (= RCCL, one GPU per rank) cannot run on a one-GPU box; what CAN be shown is that the two
    backends differ in exactly three things, all held by shard.ControlPlane — the device a rank takes, where collective tensors
    live, and init_process_group's arguments — and that bench.py reads the backend nowhere else."""
    import torch
    from reve_amd import shard
    dev = torch.device("cuda", 3)
    n = shard.control_plane("nccl", 3, 8)
    g = shard.control_plane("gloo", 3, 8)
    assert (n.backend, n.device_index, n.collective_device(dev), n.init_kwargs(dev)) == ("nccl", 3, dev, {"device_id": dev})
    assert (g.backend, g.device_index, g.collective_device(dev), g.init_kwargs(dev)) == ("gloo", 3, torch.device("cpu"), {})
    assert shard.control_plane("gloo", 5, 1).device_index == 0 and shard.control_plane("gloo", 9, 8).device_index == 1     # ranks share devices
    for bad in (lambda: shard.control_plane("nccl", 1, 1), lambda: shard.control_plane("mpi", 0, 1), lambda: shard.control_plane("gloo", 0, 0)):
        with pytest.raises(ValueError):
            bad()
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    # after the plane is built the backend's name is never consulted again: every use goes through `plane`
    after = main[main.index("plane = shard.control_plane("):]
    assert "backend ==" not in after and 'backend !=' not in after and '"gloo"' not in after and '"nccl"' not in after, "bench.py branches on the backend outside shard.control_plane"
    assert "dist.init_process_group(plane.backend, **plane.init_kwargs(dev))" in after and "cdev = plane.collective_device(dev)" in after
