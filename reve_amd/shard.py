"""Frame sharding for multi-GPU runs (one process per GPU, torch.distributed; RCCL on GPUs).

Frames of a segment are independent units (no temporal state in realesr-animevideov3, and
reve passes only directories to the upscaler, reve-shared/src/lib.rs:134-147), so the path shards
with NO data-path collective: rank r of G takes frames r, r+G, r+2G, ... of each segment, which
keeps every GPU busy inside a segment and preserves reve's segment-granular resume
(reve-cli/src/main.rs:340-343).  The only exchange is a one-off broadcast of the model bytes.

Segment arithmetic restates Video::new (reve-shared/src/lib.rs:59-86) without its off-by-one
(lib.rs:282-289 drops a frame from the last segment).
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class Segment:
    index: int
    start: int   # first frame (0-based) of the segment in the stream
    size: int


def segments(frame_count: int, segment_size: int) -> list[Segment]:
    if segment_size <= 0:
        raise ValueError("segment_size must be positive")
    out, start, i = [], 0, 0
    while start < frame_count:
        n = min(segment_size, frame_count - start)
        out.append(Segment(i, start, n))
        start += n
        i += 1
    return out


def frames_for_rank(n_frames: int, rank: int, world: int) -> list[int]:
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    return list(range(rank, n_frames, world))


@dataclass(frozen=True)
class ControlPlane:
    """Everything in which a multi-process run over RCCL differs from its dry run over gloo (bench.py, tests): the backend name,
    the device a rank takes, where collective tensors live and what init_process_group is given.  The data path — which frames a
    rank upscales, how it times them — never looks at the backend."""
    backend: str            # "nccl" (= RCCL on ROCm) | "gloo"
    device_index: int       # HIP device of this rank

    def collective_device(self, dev):
        """tensors of broadcast / all_reduce: on the rank's GPU over RCCL (xGMI), on the host over gloo"""
        import torch
        return dev if self.backend == "nccl" else torch.device("cpu")

    def init_kwargs(self, dev) -> dict:
        return {"device_id": dev} if self.backend == "nccl" else {}


def control_plane(backend: str, local_rank: int, device_count: int) -> ControlPlane:
    """nccl: rank r of the node on device r (one process per GPU; more ranks than devices is an error, not a wrap-around);
    gloo: a dry-run aid for boxes with fewer GPUs than ranks — ranks share devices, local_rank % device_count."""
    if backend not in ("nccl", "gloo"):
        raise ValueError(f"backend {backend!r}: nccl (RCCL) or gloo")
    if device_count < 1:
        raise ValueError("no HIP device visible")
    if backend == "nccl":
        if local_rank >= device_count:
            raise ValueError(f"local rank {local_rank} but only {device_count} device(s): RCCL needs one GPU per rank")
        return ControlPlane("nccl", local_rank)
    return ControlPlane("gloo", local_rank % device_count)


def _dist():
    import torch.distributed as dist
    return dist


def broadcast_model(param: bytes | None, binb: bytes | None, src: int = 0, device=None) -> tuple[bytes, bytes]:
    """Broadcast of the two ncnn model blobs from `src` (over RCCL/xGMI when `device` is a GPU)."""
    import torch
    dist = _dist()
    dev = device if device is not None else "cpu"
    lens = torch.tensor([len(param) if param is not None else 0, len(binb) if binb is not None else 0],
                        dtype=torch.int64, device=dev)
    dist.broadcast(lens, src=src)
    out = []
    for blob, n in zip((param, binb), lens.tolist()):
        if dist.get_rank() == src:
            t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
        else:
            t = torch.empty(int(n), dtype=torch.uint8, device=dev)
        dist.broadcast(t, src=src)
        out.append(t.cpu().numpy().tobytes())
    return out[0], out[1]


def gather_results(mine: dict, n_frames: int, dst: int = 0):
    """Test/diagnostic helper: collects {frame index: array} from all ranks on `dst`, in frame order."""
    dist = _dist()
    objs = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(mine, objs, dst=dst)
    if dist.get_rank() != dst:
        return None
    merged = {}
    for o in objs:
        merged.update(o)
    return [merged[i] for i in range(n_frames)]


def all_reduce_sum(x: float, device=None) -> float:
    import torch
    t = torch.tensor([x], dtype=torch.float64, device=device if device is not None else "cpu")
    _dist().all_reduce(t)
    return float(t.item())


def all_reduce_max(x: float, device=None) -> float:
    import torch
    dist = _dist()
    t = torch.tensor([x], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
