"""GPU: the torch.distributed / RCCL half of the multi-GPU design, EXECUTED — with the one GPU a test box has.

Every multi-rank run of the earlier rounds went over gloo (ranks sharing the one device); the code a real 8-GPU launch takes —
`init_process_group("nccl", device_id=dev)`, `shard.broadcast_model` / `all_reduce_*` on GPU tensors, `barrier`,
`destroy_process_group` — had never run.  A group of ONE rank over RCCL takes all of it: the communicator is created, the
collectives are enqueued on the device and complete.  What stays unmeasured is more than one device.
The ranks are started by torch.distributed.run as children of this process, before they touch the GPU.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun(script_and_args, env=None, timeout=300):       # (15-30 s when healthy; a wedged rendezvous must not eat the suite's time budget)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_and_args
    clean = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=dict(clean, **(env or {})))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_one_rank_over_real_rccl():
    d = _torchrun([os.path.join(ROOT, "tests", "workers", "rccl_rank.py")])
    assert d["backend"] == "nccl" and d["world"] == 1 and d["rank"] == 0 and d["rccl"].count(".") == 2
    assert d["broadcast_ok"] and d["bytes"] > 1_000_000            # the ~1.3 MB of model bytes went through ncclBroadcast on device tensors
    assert d["max"] == 3.5 and d["sum"] == 2.0 and d["gathered"] == [{"rank": 0, "device": 0}]
    assert d["max_lsb"] <= 1 and d["destroyed"] is True


def test_bench_takes_every_multi_rank_branch_over_rccl():
    """`bench.py --gpus 1` under torchrun with REVE_BENCH_FORCE_DIST=1: the process group over RCCL, the model broadcast, the
    all-reduces around both timed regions, the barriers and the per-rank gather all run (backend nccl, collective tensors on the
    GPU) — and cost nothing measurable: the line's `value` is that of the plain N = 1 run on the same box."""
    common = ["--gpus", "1", "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-configs", "--no-options-leg", "--min-timed-s", "3"]
    forced = _torchrun([os.path.join(ROOT, "bench.py")] + common, env={"REVE_BENCH_FORCE_DIST": "1"})
    from tests.test_gpu_parity import _bench
    plain = _bench(common)
    assert forced["n_gpus"] == 1 and forced["config"]["workload"].startswith("C2:") and forced["scaling"] == "weak"
    pr = forced["per_rank"]
    assert len(pr) == 1 and pr[0]["rank"] == 0 and pr[0]["backend"] == "nccl" and pr[0]["bcast_ms"] > 0 and pr[0]["device"] == 0
    assert forced["slowest_rank"] == {"rank": 0, "fps": pr[0]["fps"], "of_mean": 1.0} and forced["host_pinned_GBps"] > 0
    assert "per_rank" not in plain
    # (same box, back to back; the pool's box-to-box spread is +-2-4 %, run-to-run on one box is well under 1 %: -0.25 % and +0.3 % measured.
    # One more plain run if a clock ramp of the fresh box got between the two)
    def close(a, b):
        return abs(a["value"] - b["value"]) < 0.02 * b["value"] and abs(a["pipeline_fps"] - b["pipeline_fps"]) < 0.03 * b["pipeline_fps"]
    if not close(forced, plain):
        plain = _bench(common)
    assert close(forced, plain), (forced["value"], plain["value"], forced["pipeline_fps"], plain["pipeline_fps"])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump({"forced_dist_over_rccl": forced, "plain": plain}, open(os.path.join(ROOT, "gpurun_out", "bench_one_rank_rccl.json"), "w"), indent=1)
