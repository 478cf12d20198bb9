"""Worker of tests/test_rccl_rank.py: ONE rank of bench.py's multi-process path over real RCCL (backend "nccl"), started by
torch.distributed.run before anything touched the GPU.  Every call the N > 1 run makes outside the data path, on GPU tensors:
shard.control_plane -> init_process_group(device_id) -> broadcast_model -> all_reduce max / sum -> barrier -> all_gather_object ->
destroy_process_group; then the model bytes that came out of the broadcast upscale a frame.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

from reve_amd import ncnn_io, shard, synth
from reve_amd.upscaler import Upscaler


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    plane = shard.control_plane("nccl", local, torch.cuda.device_count())
    torch.cuda.set_device(plane.device_index)
    dev = torch.device("cuda", plane.device_index)
    cdev = plane.collective_device(dev)
    assert cdev == dev and plane.init_kwargs(dev) == {"device_id": dev}
    dist.init_process_group(plane.backend, **plane.init_kwargs(dev))
    out = {"rank": rank, "world": world, "backend": dist.get_backend(), "rccl": ".".join(str(x) for x in torch.cuda.nccl.version())}
    w = synth.make_weights(2)
    param, binb = ncnn_io.build_param_text(2).encode(), ncnn_io.build_bin(w)
    p2, b2 = shard.broadcast_model(param if rank == 0 else None, binb if rank == 0 else None, src=0, device=cdev)
    out["broadcast_ok"] = (p2 == param and b2 == binb)
    out["bytes"] = len(p2) + len(b2)
    out["max"] = shard.all_reduce_max(3.5 + rank, device=cdev)
    out["sum"] = shard.all_reduce_sum(2.0, device=cdev)
    dist.barrier()
    gathered = [None] * world
    dist.all_gather_object(gathered, {"rank": rank, "device": plane.device_index})
    out["gathered"] = gathered
    # the bytes that travelled are a model: one frame through the HIP path against the oracle
    from oracle import ref
    img = synth.noise_frame(5, 96, 64)
    with Upscaler(2, param=p2, bin=b2, device=plane.device_index) as up:
        d = np.abs(up.upscale(img).astype(np.int32) - ref.upscale(w, img).astype(np.int32))
    out["max_lsb"] = int(d.max())
    dist.barrier()
    dist.destroy_process_group()
    out["destroyed"] = not dist.is_initialized()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
