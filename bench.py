#!/usr/bin/env python3
"""bench.py — upscaled frames/sec of the realesr-animevideov3 path on N MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one synthetic frame: conv_first -> 16 body convs ->
conv_last (+pixel-shuffle, residual, post-process), u8 RGB in HBM -> u8 RGB in HBM, called through
the C ABI (reve_upscale_rgb8_device).  Workload = BASELINE config 2: 1920x1080 -> 3840x2160, x2,
S-noise frames (seed 0x5EED0001), synthetic weights of the real architecture (no model files or
datasets exist offline).  Frames are resident in HBM when the timed region starts; the
PCIe-inclusive rate of the submit/wait ring is reported separately (`pipeline_fps`).

Multi-GPU: one process per GPU, frames sharded with no data-path collective (weak scaling: every
rank upscales K steps); the only exchange is the RCCL broadcast of the model bytes from rank 0.
`--gpus N` without a torchrun environment starts the N ranks itself (child processes, before this
process has touched a GPU) and exits with their status; a rank count that does not match --gpus is an error.
With N > 1 the default workload is BASELINE config 4's SCHEDULE on the same frames (round 6): the stream
(K x frames_per_step x N frames) in segments of --segmentsize 1000, rank r takes frames r, r+N, ... of every
segment and completes each segment before the next (reve's resume granularity,
reve-cli/src/main.rs:340-343); N = 1 stays C2 — the same frames without the segment fences (the N = 1
line's `configs.C4_1gpu` is config 4's schedule on one GPU: within 1 % of C2).  The N > 1 line carries
`per_rank` (each rank's own frames/s in HBM and through the ring, the CPUs it was bound to, its GPU's
local_cpulist, what its pinned allocations and the model broadcast took), `slowest_rank` and
`host_pinned_GBps` (aggregate H2D + D2H through the pinned rings) — what one needs to diagnose a slow rank.
REVE_BENCH_FORCE_DIST=1 (tests) makes a one-rank run under torchrun take every one of those branches over RCCL.

--steps K is honoured exactly, but a step is a BATCH of `frames_per_step` frames sized so that the timed
region lasts at least --min-timed-s (5 s: K = 20 would otherwise time 47 ms, and a one-second region is too short for
the driver's own samplers to corroborate); the line reports steps, frames_per_step and both per-step and per-frame
times.  --workload C4 without --steps walks the literal 8000-frame stream (strong scaling).

Evaluation (round 6): the library's default — option "winograd" = auto — is what the headline times: the body pairs by
Winograd F(2,3) along the row when the weights' conditioning estimate kappa is under 0.5 (it is 0.017 for the synthetic
weights), the direct sums otherwise; `roofline.evaluation` / `roofline.kappa` say which and why, in the headline and in every
`configs` leg, and `option_direct` is the same frames with the direct kernels pinned (--winograd 0), so that both numbers are in
every line.  The roofline prices ALGORITHMIC flops whatever the evaluation.

Besides the HBM-resident headline (`value`, as the bench contract defines it) the line carries
  * `pipeline_fps`: the same number of frames from pinned host memory through the reve_submit/reve_wait ring
    (hipMemcpyAsync H2D, kernel chain, D2H on three streams — the pipeline north_star names), with per-stage times,
    overlap efficiency and the PCIe bound, and the per-kernel split of a frame;
  * `configs` (N = 1, headline workload C2): the OTHER configurations of BASELINE.json timed in the same process on the same
    box, one short leg each (>= --leg-timed-s of frames in HBM, then as long through the ring): `C2_tile200` — the mode an
    unmodified reve gets (the binary tiles at 200 px with a 10-px apron; reve passes no -t, reve-shared/src/lib.rs:134-147)
    — `C3` (1080p x4), `C3_literal` (960x540 x4), `C5` (4K x2) and `C4_1gpu` (the segmented 1080p stream of config 4 on this one GPU,
    three 1000-frame segments), each with value, roofline, launch_us, pipeline_fps, pcie_bound_fps and slowest_stage.  Informational: `value`, `metric` and `config.workload` stay C2's.
Secondary figures are derived, not typed: `roofline.mfma_flop_executed` from the launch geometry the library reports
(option "pair_mfma_per_launch": strips x segments x steps x waves x MFMAs per step) and `roofline.traffic` from
profiles/traffic.json (the entry of the kernel that ran: k_wino or k_pair), which records the sha256 of the kernel sources it was
measured on — `traffic_stale` says whether the library timed here was built from the same ones (reve_build_info()).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

from reve_amd import ncnn_io, shard, synth
from reve_amd.hostcpus import usable_cpus
from reve_amd.upscaler import Upscaler, pinned_array, free_pinned

FLOP_PER_LR_PX = {2: 1196928, 3: 1214208, 4: 1238400}    # 2*MAC of the 18 convs
BODY_FLOP_PER_LR_PX = 2 * 36864                           # one 64->64 3x3 layer
MFMA_FLOP = 2 * 16 * 16 * 32                              # one v_mfma_f32_16x16x32_f16
PEAK_F16_MFMA_TFLOPS = 2500.0                             # dense, MI355X_MICROARCH.md chip table
RING = 16                                                 # distinct frames cycled
MIN_TIMED_S = 5.0                                         # the headline's timed region lasts at least this long
LEG_TIMED_S = 1.5                                         # ... and each leg of `configs`
NAMED = {"C2": (1920, 1080, 2), "C3": (1920, 1080, 4), "C3-literal": (960, 540, 4), "C4": (1920, 1080, 2), "C5": (3840, 2160, 2)}
# the other BASELINE configurations, timed after the headline at N = 1: key -> (workload, tile)
# (C4 on the one GPU: the segmented stream — every 1000-frame segment completed before the next, reve's resume granularity — long
# enough for three segments; its 8-GPU sharding is the N > 1 run's to measure)
LEGS = (("C2_tile200", "C2", 200, None), ("C3", "C3", 0, None), ("C3_literal", "C3-literal", 0, None), ("C5", "C5", 0, None), ("C4_1gpu", "C4", 0, 4.2))


def cpu_baseline(weights, frame, w, h):
    """The oracle (CPU restatement standing in for the ncnn CPU path; `realesrgan-ncnn -g -1` does
    not exist on this box) on a bounded sample of the same workload: a 640x360 crop (1/9 of one C2
    frame) first; if that predicts a whole frame in under ~30 s, whole 1920x1080 frames for ~10 s."""
    from oracle import ref
    threads, visible = usable_cpus()              # one OpenMP thread per CPU the process is allowed to use
    crop = np.ascontiguousarray(frame[:360, :640])
    ref.upscale(weights, crop[:64, :64], nthreads=threads)          # warm-up (page-in, thread pool)
    t0 = time.perf_counter()
    ref.upscale(weights, crop, nthreads=threads)
    dt = time.perf_counter() - t0
    frac, what = crop.shape[0] * crop.shape[1] / float(w * h), f"640x360 crop (1/9 of one {w}x{h} S-noise frame)"
    if dt / frac < 30.0:
        # whole frames until about 10 s of CPU work have been timed (at most 8 frames)
        n, t0 = 0, time.perf_counter()
        while n < 8 and (n == 0 or time.perf_counter() - t0 < 10.0):
            ref.upscale(weights, frame, nthreads=threads)
            n += 1
        dt = time.perf_counter() - t0
        frac, what = float(n), f"{n} whole {w}x{h} S-noise frame(s)"
    return {"value": round(frac / dt, 4), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{what} x2, fp16-storage mode, {dt:.1f} s of wall time on {threads} OpenMP threads (the process may use {threads} "
                      f"of the {visible} CPUs it sees); CPU restatement (oracle) standing in for the ncnn CPU path"}


def self_launch(args, argv):
    """`bench.py --gpus N` outside torchrun: start the N ranks as children of this process, which has not
    touched a GPU (torch.cuda.device_count() does not initialise one), and leave with their exit status."""
    backend = os.environ.get("REVE_BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count()
    if have == 0:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    if backend == "nccl" and have < args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but only {have} device(s) visible (REVE_BENCH_BACKEND=gloo shares devices for a dry run)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd)


def traffic_record(lpl, whole_1080p, winograd):
    """PMC-derived HBM bytes per body launch (profiles/traffic.json, written by scripts/install_profiles.py from rocprofv3 --pmc
    passes) with its provenance: the record names the sha256 of the kernel sources of the library it was measured on;
    `stale` = the library timed here was built from different ones (or the record predates the stamp).  One entry per kernel:
    `wino_*` for the Winograd pairs, `pair_*` for the direct pairs, `body_*` for one layer per launch."""
    from reve_amd import _lib
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not (os.path.exists(tpath) and whole_1080p):          # collected on whole-frame 1080p body launches
        return None, None, None
    tj = json.load(open(tpath))
    which = ("wino" if winograd else "pair") if lpl == 2 else "body"
    traffic = tj.get(f"{which}_hbm_bytes_per_launch")
    if traffic is None:
        return None, None, None
    sha_key = "wino_src_sha256" if which == "wino" else "pair_src_sha256"
    built = _lib.build_info().get(sha_key)
    stale = not (built and tj.get(sha_key) == built)
    return traffic, f"profiles/traffic.json[{which}_hbm_bytes_per_launch] ({tj.get('source', 'rocprofv3 --pmc passes')}); not measured in this run", stale


def parse_workload(name):
    """C2 / C3 / C3-literal / C4 / C5, or WxH[xS] -> (W, H, scale)"""
    if name in NAMED:
        return NAMED[name]
    try:
        dims = [int(x) for x in name.lower().split("x")]
        W, H, S = dims[0], dims[1], (dims[2] if len(dims) > 2 else 2)
        assert len(dims) in (2, 3) and W > 0 and H > 0 and S in (2, 3, 4)
        return W, H, S
    except Exception:
        raise SystemExit(f"--workload {name}: not C2 / C3 / C3-literal / C4 / C5 and not WxH[xS]")


class Leg:
    """One workload on one context: frames of W x H resident in HBM through reve_upscale_rgb8_device(_batch), then the same
    count from pinned host memory through the reve_submit / reve_wait ring."""

    def __init__(self, args, workload, tile, rank, world, local, dev, cdev, param, binb, scale_of_model):
        self.args, self.workload, self.tile = args, workload, tile
        self.rank, self.world, self.dev, self.cdev = rank, world, dev, cdev
        self.dist_on = dist.is_initialized()          # (world > 1, or a one-rank group under REVE_BENCH_FORCE_DIST=1)
        self.W, self.H, self.S = parse_workload(workload)
        assert scale_of_model == self.S
        W, H, S = self.W, self.H, self.S
        self.up = up = Upscaler(S, param=param, bin=binb, device=local, tile=tile)
        if args.fuse != "auto":
            up.set_option("fuse_pairs", int(args.fuse))
        if args.batch != "auto":
            up.set_option("batch", int(args.batch))
        self.wino_mode = {"0": 0, "1": 1, "auto": 2}[args.winograd]
        up.set_option("winograd", self.wino_mode)
        self.winograd = bool(up.get_option("winograd"))            # the evaluation in force (auto: the library's rule on these weights)
        self.kappa = up.get_option("winograd_kappa_permille") / 1000.0
        # synthetic stream: rank r owns frames r, r+G, ... of the stream; a ring of 16 lives in HBM
        gen = {"noise": synth.noise_frame, "toon": synth.toon_frame, "video": synth.video_frame}[args.frames]
        self.frames_np = [gen(rank + i * world, W, H) for i in range(RING)]
        self.src = [torch.from_numpy(f).to(dev) for f in self.frames_np]
        probe = torch.empty((H * S, W * S, 3), dtype=torch.uint8, device=dev)
        up.upscale_device(self.src[0].data_ptr(), W, H, probe.data_ptr())      # (lays the geometry out: how many frames share a launch)
        up.sync()
        self.bf = up.get_option("batch_frames")          # frames that share a kernel chain at this size (1: every frame has its own)
        self.dst = [probe] + [torch.empty_like(probe) for _ in range(max(2, self.bf) - 1)]
        torch.cuda.synchronize()

    def close(self):
        self.up.close()
        self.src = self.dst = None
        torch.cuda.empty_cache()

    def frames(self, i0, n):
        """frames i0 .. i0 + n - 1 of this rank's share; small frames go through the chain bf at a time"""
        up, src, dst, W, H, bf = self.up, self.src, self.dst, self.W, self.H, self.bf
        if bf == 1:
            for i in range(i0, i0 + n):
                up.upscale_device(src[i % RING].data_ptr(), W, H, dst[i & 1].data_ptr())
            return
        for j0 in range(0, n, bf):
            k = min(bf, n - j0)
            up.upscale_device_batch([src[(i0 + j0 + j) % RING].data_ptr() for j in range(k)], [dst[j].data_ptr() for j in range(k)], W, H)

    def fence(self):
        self.up.sync()
        torch.cuda.synchronize()
        if self.dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    def run(self, steps, warmup, min_timed_s, pcie=True, direct_leg=False, schedule=None):
        """warm-up, then EXACTLY `steps` timed steps of `frames_per_step` frames each (sized so that the region lasts
        >= min_timed_s), bracketed by barrier + synchronize; then the pipeline over the same frame count.  `schedule` "C4": the
        stream in segments, each completed before the next (default: the workload's own).  Returns the pieces of a line."""
        args, up, bf, world, cdev, W, H, S = self.args, self.up, self.bf, self.world, self.cdev, self.W, self.H, self.S
        dist_on = self.dist_on
        segmented = (schedule or self.workload) == "C4"
        self.frames(0, max(warmup, bf))
        self.fence()                      # arenas allocated, kernels loaded: what follows is steady state
        t0 = time.perf_counter()
        n_cal = 12 * bf
        self.frames(0, n_cal)
        up.sync()
        per_frame_s = (time.perf_counter() - t0) / n_cal / 1.1     # margin: better a little over min_timed_s than under
        fps_step = max(1, math.ceil(min_timed_s / (steps * per_frame_s)))
        fps_step = (fps_step + bf - 1) // bf * bf          # whole batches
        if dist_on:
            fps_step = int(round(shard.all_reduce_max(float(fps_step), device=cdev)))
        n_frames = steps * fps_step                 # per rank
        # C4: the stream (n_frames x world frames) in segments; this rank's share of segment s is frames r, r+G, ... of it
        seg_sizes = None
        if segmented:
            segs = shard.segments(n_frames * world, args.segmentsize)
            seg_sizes = [len(shard.frames_for_rank(sg.size, self.rank, world)) for sg in segs]
            assert sum(seg_sizes) == n_frames or world > 1
            n_frames = sum(seg_sizes)
        up.set_profiling(True)
        up.reset_stats()

        self.fence()
        t0 = time.perf_counter()
        if seg_sizes is None:
            self.frames(0, n_frames)
        else:
            i = 0
            for n_seg in seg_sizes:
                self.frames(i, n_seg)
                i += n_seg
                up.sync()     # segment complete: where reve rewrites video.temp (main.rs:340-343)
        up.sync()
        own_elapsed = time.perf_counter() - t0      # this rank's own frames done (before it waits for the others)
        self.fence()
        elapsed = time.perf_counter() - t0
        if dist_on:
            elapsed = shard.all_reduce_max(elapsed, device=cdev)
            total_frames = int(round(shard.all_reduce_sum(float(n_frames), device=cdev)))
        else:
            total_frames = n_frames
        st = up.stats()
        lpl = max(int(st["body_layers_per_launch"]), 1)
        mfma_per_launch = up.get_option("pair_mfma_per_launch") if lpl == 2 else None
        geometry = {k: up.get_option("pair_" + k) for k in ("strips", "segments", "seg_rows", "units")} if lpl == 2 else None

        # ---- informational: the same frames with the direct pair kernels pinned (library option "winograd" = 0: what round 5
        # shipped as the default, and what REVE_WINOGRAD=0 gives), so that the line of any box carries both numbers.  N = 1 only,
        # after the timed region, never part of `value`.
        direct = None
        if direct_leg and self.winograd:
            up.set_profiling(False)
            up.set_option("winograd", 0)
            n_w = min(n_frames, max(bf, 300 // bf * bf))
            self.frames(0, max(8, bf))
            self.fence()
            tw = time.perf_counter()
            self.frames(0, n_w)
            self.fence()
            tw = time.perf_counter() - tw
            up.set_option("winograd", self.wino_mode)
            up.set_profiling(True)
            direct = {"value": round(n_w / tw, 2), "unit": "frames/s", "frames": n_w, "evaluation": "direct",
                      "roofline_frac_whole_path": round(n_w / tw * FLOP_PER_LR_PX[S] * W * H / (PEAK_F16_MFMA_TFLOPS * 1e12), 4),
                      "note": "same frames, body pairs by the direct sums (k_pair; option \"winograd\" = 0 / REVE_WINOGRAD=0): the round-5 default; informational"}

        # ---- the pipeline north_star names ("in-process reve_submit/wait, ring depth >= 3"): the same frames from
        # pinned host memory through hipMemcpyAsync H2D -> kernel chain -> D2H on three streams, over the SAME frame count as the
        # HBM-resident region above (every rank at once).  Reported as `pipeline_fps`; `value` stays the HBM-resident rate because
        # the bench contract defines it so (inputs resident in HBM when the timed region starts; a PCIe-inclusive rate is never `value`).
        pipe_fps = ring = None
        own_pipe_s = pinned_ms = None
        if pcie:
            n = n_frames
            depth = 3 if bf == 1 else 2 * bf          # (frames that share launches: a batch computing and a batch filling)
            tp = time.perf_counter()
            hin = [pinned_array((H, W, 3)) for _ in range(depth)]
            hout = [pinned_array((H * S, W * S, 3)) for _ in range(depth)]
            for k in range(depth):
                hin[k][...] = self.frames_np[k % RING]
            pinned_ms = (time.perf_counter() - tp) * 1e3          # allocation + first touch of the ring's pinned frames
            for i in range(depth):          # warm the ring's device slots
                up.submit(i, hin[i], hout[i])
            for _ in range(depth):
                up.wait()
            up.reset_stats()
            self.fence()
            # (config 4's schedule: the ring drains at every segment end too — a segment is complete when its last frame is back on the host)
            seg_ends = set(np.cumsum(seg_sizes).tolist()) if seg_sizes is not None else set()
            t1 = time.perf_counter()
            inflight = 0
            for i in range(n):
                if inflight >= depth:
                    up.wait()
                    inflight -= 1
                up.submit(i, hin[i % depth], hout[i % depth])
                inflight += 1
                if i + 1 in seg_ends:
                    while inflight:
                        up.wait()
                        inflight -= 1
            while inflight:
                up.wait()
                inflight -= 1
            own_pipe_s = dt = time.perf_counter() - t1
            if dist_on:
                dt = shard.all_reduce_max(dt, device=cdev)
            pipe_fps = total_frames / dt
            rs = up.stats()
            if rs["ring_frames"]:
                k = rs["ring_frames"]
                stage = {"h2d": rs["h2d_ms_total"] / k, "chain": rs["chain_ms_total"] / k, "d2h": rs["d2h_ms_total"] / k}
                ring = {"frames": int(k), "ring_depth": depth, "timed_s": round(dt, 3),
                        "h2d_ms": round(stage["h2d"], 4), "chain_ms": round(stage["chain"], 4),
                        "d2h_ms": round(stage["d2h"], 4), "wall_ms_per_frame": round(rs["ring_wall_ms"] / k, 4),
                        # 1.0 = the ring runs at the speed of its slowest stage (the other two fully hidden under it)
                        "overlap_efficiency": round(max(stage.values()) * k / rs["ring_wall_ms"], 4) if rs["ring_wall_ms"] > 0 else None,
                        "slowest_stage": max(stage, key=stage.get),
                        # what the PCIe link alone would allow per GPU (uploads and downloads run on separate copy engines): the cap
                        # on any kernel gain; at x4 the 99.5 MB download is within 20 % of the chain's time
                        "pcie_bound_fps": round(1e3 / max(stage["h2d"], stage["d2h"]), 1) if max(stage["h2d"], stage["d2h"]) > 0 else None,
                        # host traffic of this rank's ring: every frame crosses the link once each way
                        "host_pinned_GBps": round(n * (W * H * 3) * (1 + S * S) / own_pipe_s / 1e9, 2)}
            for a in hin + hout:
                free_pinned(a)
        up.set_profiling(False)

        # the dominant kernel: one body launch = `lpl` 64->64 layers (1: k_body, 2: the fused pair k_pair / k_wino); the library's
        # events bracket the 16 layers of a frame and count layers, so the launch time is the per-layer time x lpl
        body_ms = st["body_ms_total"] / max(st["body_launches"], 1) * lpl
        body_flop = BODY_FLOP_PER_LR_PX * W * H * lpl * bf
        achieved = body_flop / (body_ms * 1e-3) / 1e12 if body_ms > 0 else 0.0
        fps = total_frames / elapsed
        kt = max(st["frames_timed"], 1)
        kernel = ("k_wino (two 64->64 3x3 conv + bias + PReLU layers per launch by Winograd F(2,3) along the row)" if self.winograd else
                  "k_pair (two 64->64 3x3 conv + bias + PReLU layers per launch, the layer between them in LDS)") if lpl == 2 else "k_body (64->64 3x3 conv + bias + PReLU)"
        wino_ran = self.winograd and lpl == 2
        how = {0: "pinned (--winograd 0)", 1: "forced (--winograd 1)", 2: f"auto: kappa {self.kappa:.3f} {'<' if self.winograd else '>='} 0.5"}[self.wino_mode]
        roofline = {"bound": "mfma", "kernel": kernel, "achieved": round(achieved, 1), "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_F16_MFMA_TFLOPS, 4), "launch_us": round(body_ms * 1e3, 2),
                    "launches_timed": st["body_launches"] // lpl, "layers_per_launch": lpl, "frames_per_launch": bf,
                    "algorithmic_flop_per_launch": body_flop,
                    # what the matrix cores execute for it (the roofline above prices the ALGORITHMIC flops whatever the evaluation):
                    # MFMA instructions of one launch as the library lays it out — strips x segments x (steps of both layers) x
                    # 2 waves x 288 per step (192: Winograd) — x 16,384 FLOP; direct sums = algorithmic + the strips' recomputed
                    # columns and the segments' halo rows.  profiles/rNN/pmc_summary.json: SQ_VALU_MFMA_BUSY_CYCLES / 16 is the same count
                    "mfma_flop_executed": mfma_per_launch * MFMA_FLOP if mfma_per_launch else None,
                    "mfma_instructions_per_launch": mfma_per_launch, "launch_geometry": geometry,
                    # the evaluation that ran, and the conditioning estimate of the loaded weights the library's rule compared
                    # (reve_get_option "winograd", "winograd_kappa_permille")
                    "evaluation": f"winograd F(2,3) along the row ({how})" if wino_ran else f"direct ({how})",
                    "winograd": bool(wino_ran), "kappa": round(self.kappa, 4), "kappa_limit": 0.5}
        mine = {"rank": self.rank, "frames": n_frames, "fps": round(n_frames / own_elapsed, 2), "own_timed_s": round(own_elapsed, 3),
                "pipeline_fps": round(n_frames / own_pipe_s, 2) if own_pipe_s else None,
                "host_pinned_GBps": ring["host_pinned_GBps"] if ring else None,
                "pinned_alloc_ms": round(pinned_ms, 1) if pinned_ms is not None else None,
                "chain_ms": ring["chain_ms"] if ring else None, "h2d_ms": ring["h2d_ms"] if ring else None, "d2h_ms": ring["d2h_ms"] if ring else None,
                "launch_us": roofline["launch_us"]}
        return {"fps": fps, "elapsed": elapsed, "n_frames": n_frames, "total_frames": total_frames, "fps_step": fps_step, "seg_sizes": seg_sizes,
                "roofline": roofline, "body_ms": body_ms, "lpl": lpl, "pipe_fps": pipe_fps, "ring": ring, "direct": direct, "mine": mine,
                "whole_path_frac": round(fps / world * FLOP_PER_LR_PX[S] * W * H / (PEAK_F16_MFMA_TFLOPS * 1e12), 4),
                "stages_ms": {"conv_first": round(st["first_ms_total"] / kt, 4), "body_x16": round(st["body_ms_total"] / kt, 4),
                              "conv_last": round(st["last_ms_total"] / kt, 4), "chain": round(st["frame_ms_total"] / kt, 4),
                              "frames_timed": st["frames_timed"]}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 1000; --workload C4 without --steps: the literal 8000-frame stream, 8000 / N frames per rank)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pcie", action="store_true", help="skip the host<->device submit/wait ring leg")
    ap.add_argument("--no-options-leg", action="store_true", help="skip the informational leg with the direct kernels pinned (`option_direct`)")
    ap.add_argument("--no-configs", action="store_true", help="skip the legs over the other BASELINE configurations (`configs`)")
    ap.add_argument("--pcie", action="store_true", help="(default now; kept for old command lines)")
    ap.add_argument("--tile", type=int, default=0, help="0 = whole frame (default, the headline run); N = ncnn-compat tiling")
    ap.add_argument("--frames", default="noise", choices=["noise", "toon", "video"],
                    help="synthetic content: uniform noise (default; the worst case for the power-capped MFMA pipe), flat-shaded toon "
                         "frames, or `video`: toon + the +-2 LSB grain and 8x8 block edges of a decoded H.264 frame (the model's real input)")
    ap.add_argument("--workload", default=None,
                    help="BASELINE.json config to run.  Default: C2 at N = 1 (1080p x2: the headline metric's workload) and config 4's "
                         "schedule at N > 1 (the same frames as a stream in segments of --segmentsize, frame-sharded over the ranks, each segment "
                         "completed before the next).  C3, C3-literal, C4, C5; or a frame size WxH[xS] — "
                         "e.g. 640x480, 256x256 (BASELINE config 1's shape), 100x100: the sizes of the reference's own assets "
                         "(reve-cli/assets/), which go through the kernel chain several frames per launch")
    ap.add_argument("--batch", default="auto", choices=["auto", "0", "1"], help="small frames several per launch (library option \"batch\")")
    ap.add_argument("--winograd", default="auto", choices=["0", "1", "auto"],
                    help="library option \"winograd\": auto (the library's default: Winograd pairs iff the weights' conditioning allows), 0 (direct kernels pinned), 1 (forced)")
    ap.add_argument("--segmentsize", type=int, default=1000, help="C4: frames per segment (reve's default, lib.rs:228)")
    ap.add_argument("--min-timed-s", type=float, default=MIN_TIMED_S)
    ap.add_argument("--leg-timed-s", type=float, default=LEG_TIMED_S)
    ap.add_argument("--fuse", default="auto", choices=["auto", "0", "1"],
                    help="body layers two per launch (kernels_pair.hip): auto = the library's default for the geometry")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    if args.workload is not None:
        parse_workload(args.workload)          # (a bad name stops here, before any rank is started)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the line would report the wrong n_gpus")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    default_workload = args.workload is None
    if default_workload:
        args.workload = "C2" if world == 1 else "C4"
    # The control plane (shard.control_plane): backend nccl (= RCCL) puts rank r on device r and the collective tensors on it;
    # REVE_BENCH_BACKEND=gloo is a dry-run aid for boxes with fewer GPUs than ranks — ranks share devices (local % device_count)
    # and the collectives run over gloo on the host.  Nothing else in this file depends on the backend.
    backend = os.environ.get("REVE_BENCH_BACKEND", "nccl")
    plane = shard.control_plane(backend, local, torch.cuda.device_count())
    local = plane.device_index
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # this rank's host side next to its GPU: bind the process to the CPUs of the GPU's PCIe root BEFORE the first
    # pinned allocation, so that the ring's pinned frames are first touched on that NUMA node (never beyond the mask it has)
    from reve_amd import _lib as _revelib
    affinity_before = os.sched_getaffinity(0)
    bound_cpus = _revelib.load().reve_bind_thread_to_device(local) if os.environ.get("REVE_BENCH_BIND", "1") == "1" else 0
    cdev = plane.collective_device(dev)      # where collective tensors live
    # a group of one (REVE_BENCH_FORCE_DIST=1 under torchrun --nproc-per-node 1): every branch the N > 1 run takes — process group
    # over RCCL, model broadcast, the all-reduces around the timed regions, barriers, the per-rank gather — on the one GPU of a test box
    dist_on = world > 1 or os.environ.get("REVE_BENCH_FORCE_DIST", "0") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        dist.init_process_group(plane.backend, **plane.init_kwargs(dev))     # nccl == RCCL on ROCm

    # model: rank 0 builds the ncnn files' bytes, everyone else receives them over RCCL/xGMI
    bcast_ms = [0.0]

    def model_bytes(scale):
        weights = synth.make_weights(scale) if rank == 0 else None
        param = ncnn_io.build_param_text(scale).encode() if rank == 0 else None
        binb = ncnn_io.build_bin(weights) if rank == 0 else None
        if dist_on:
            tb = time.perf_counter()
            param, binb = shard.broadcast_model(param, binb, src=0, device=cdev)
            bcast_ms[0] += (time.perf_counter() - tb) * 1e3
        return weights, param, binb

    scale = parse_workload(args.workload)[2]
    weights, param, binb = model_bytes(scale)
    leg = Leg(args, args.workload, args.tile, rank, world, local, dev, cdev, param, binb, scale)
    W, H, SCALE, bf = leg.W, leg.H, leg.S, leg.bf
    strong = args.workload == "C4" and args.steps is None
    steps = args.steps if args.steps is not None else (8000 // world if args.workload == "C4" else 1000)
    if steps < 1:
        raise SystemExit("--steps must be >= 1")
    r = leg.run(steps, args.warmup, args.min_timed_s, pcie=not args.no_pcie,
                direct_leg=world == 1 and not args.no_options_leg and args.workload != "C4")
    frames_np0 = leg.frames_np[0]
    leg.close()

    # ---- per-rank diagnostics (N > 1): every rank's own figures to rank 0
    per_rank = None
    if dist_on:
        cpulist = C_cpulist(_revelib, local)
        mine = dict(r["mine"], device=local, bound_cpus=int(bound_cpus), of_visible=len(affinity_before), local_cpulist=cpulist,
                    bcast_ms=round(bcast_ms[0], 2), backend=plane.backend)
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        per_rank = sorted(gathered, key=lambda m: m["rank"])

    # ---- the other BASELINE configurations on this box, in this process (N = 1, headline = C2 as the driver runs it)
    configs = None
    if world == 1 and not args.no_configs and args.workload == "C2" and args.tile == 0 and args.frames == "noise" and args.winograd == "auto":
        configs = {}
        models = {scale: (param, binb)}
        for key, wl, tile, timed_s in LEGS:
            s = NAMED[wl][2]
            if s not in models:
                models[s] = model_bytes(s)[1:]
            lg = Leg(args, wl, tile, rank, world, local, dev, cdev, models[s][0], models[s][1], s)
            q = lg.run(8, 6, max(args.leg_timed_s, timed_s) if timed_s else args.leg_timed_s, pcie=not args.no_pcie)
            lg.close()
            rf = q["roofline"]
            configs[key] = {"workload": f"{lg.W}x{lg.H} -> {lg.W * lg.S}x{lg.H * lg.S} x{lg.S}" + (f", the binary's tiling: {tile}-px tiles + 10-px apron" if tile else ", whole frame"),
                            "value": round(q["fps"], 2), "unit": "frames/s", "frames": q["n_frames"], "timed_s": round(q["elapsed"], 3),
                            "ms_per_frame": round(q["elapsed"] / q["n_frames"] * 1e3, 4), "frames_per_launch": lg.bf,
                            "roofline": {k: rf[k] for k in ("kernel", "achieved", "frac", "launch_us", "layers_per_launch", "algorithmic_flop_per_launch", "mfma_flop_executed",
                                                            "evaluation", "winograd", "kappa")},
                            "launch_us": rf["launch_us"], "roofline_frac_whole_path": q["whole_path_frac"], "stages_ms": q["stages_ms"],
                            "pipeline_fps": round(q["pipe_fps"], 2) if q["pipe_fps"] else None,
                            "pcie_bound_fps": q["ring"]["pcie_bound_fps"] if q["ring"] else None,
                            "slowest_stage": q["ring"]["slowest_stage"] if q["ring"] else None,
                            "overlap_efficiency": q["ring"]["overlap_efficiency"] if q["ring"] else None}
            if q["seg_sizes"] is not None:
                configs[key]["segments"] = len(q["seg_sizes"])
                configs[key]["segmentsize"] = args.segmentsize

    if rank == 0:
        n_frames, total_frames, elapsed, seg_sizes, lpl = r["n_frames"], r["total_frames"], r["elapsed"], r["seg_sizes"], r["lpl"]
        traffic, traffic_source, traffic_stale = traffic_record(lpl, (W, H) == (1920, 1080) and args.tile == 0, r["roofline"]["winograd"])
        roofline = dict(r["roofline"], traffic=traffic, traffic_source=traffic_source, traffic_stale=traffic_stale)
        body_ms = r["body_ms"]
        line = {
            "metric": "upscaled frames/sec 1080p->4K x2 realesr-animevideov3" if args.workload in ("C2", "C4")
                      else f"upscaled frames/sec {W}x{H} x{SCALE} realesr-animevideov3",
            "value": round(r["fps"], 2), "unit": "frames/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 4), "frames_per_step": n_frames / steps,
            "ms_per_frame": round(elapsed / n_frames * 1e3, 4), "timed_s": round(elapsed, 3),
            "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic" if args.frames == "noise" else f"synthetic-{args.frames}",
            "config": {"workload": f"{args.workload}: {W}x{H} -> {W * SCALE}x{H * SCALE} x{SCALE} realesr-animevideov3 (SRVGGNetCompact 64x16), "
                                   f"S-{args.frames} frames resident in HBM, synthetic weights"
                                   + (f", the stream in segments of {args.segmentsize} frames, each completed before the next (BASELINE config 4's schedule)" if seg_sizes else ""),
                       "frames_per_gpu": n_frames,
                       "frames_total": total_frames,
                       "frame_sharding": f"dp{world}, rank r takes frames r, r+{world}, ..." + (" of every segment" if seg_sizes else ""),
                       "tile": args.tile, "frames_per_launch": bf},
            "roofline_frac_whole_path": r["whole_path_frac"],
            "value_is": "frames resident in HBM (bench contract); the host-to-host pipeline of north_star is pipeline_fps",
            "roofline": roofline,
            # the same launch against the HBM roofline (layer-per-launch round-trips the activations):
            # algorithmic bytes = fp16 activations in + out
            "roofline_hbm": {"bound": "hbm", "achieved": round(2 * W * H * 128 * bf / (body_ms * 1e-3) / 1e9, 1) if body_ms > 0 else 0.0,
                             "peak": 8000.0, "unit": "GB/s",
                             "frac": round(2 * W * H * 128 * bf / (body_ms * 1e-3) / 8e12, 4) if body_ms > 0 else 0.0,
                             "traffic": traffic, "traffic_source": traffic_source, "traffic_stale": traffic_stale,
                             "algorithmic_bytes_per_launch": 2 * W * H * 128 * bf},
            # device time of one frame's kernels (HIP events on the launch stream, rank 0)
            "stages_ms": r["stages_ms"],
        }
        if seg_sizes is not None:
            line["config"]["segments"] = len(seg_sizes)
            line["config"]["segmentsize"] = args.segmentsize
        if world > 1 and default_workload:
            line["config"]["n1_is"] = ("C2: the same frames without the segment fences; config 4's schedule on one GPU is the N = 1 line's configs.C4_1gpu "
                                       "(within 1 % of C2: profiles/r05, r06)")
        if r["pipe_fps"] is not None:
            line["pipeline_fps"] = round(r["pipe_fps"], 2)              # pinned host -> H2D -> chain -> D2H -> pinned host, ring depth 3
            line["pipeline"] = r["ring"]
            line["pcie_inclusive_fps"] = round(r["pipe_fps"], 2)        # (the name rounds 1-2 used for the same figure)
            line["pcie_ring"] = r["ring"]
        if r["direct"] is not None:
            line["option_direct"] = r["direct"]
        if configs is not None:
            line["configs"] = configs
        if per_rank is not None:
            # what a slow rank looks like from rank 0: each rank's own rate in HBM and through its ring, where its host side sits
            line["per_rank"] = per_rank
            slow = min(per_rank, key=lambda m: m["fps"])
            line["slowest_rank"] = {"rank": slow["rank"], "fps": slow["fps"], "of_mean": round(slow["fps"] * len(per_rank) / sum(m["fps"] for m in per_rank), 4)}
            if all(m.get("host_pinned_GBps") is not None for m in per_rank) and r["pipe_fps"]:
                # aggregate H2D + D2H through all ranks' pinned rings at the job's pipeline rate (every frame crosses the link once each way)
                line["host_pinned_GBps"] = round(r["pipe_fps"] * (W * H * 3) * (1 + SCALE * SCALE) / 1e9, 2)
        line["host_placement"] = {"bound_cpus": int(bound_cpus), "of_visible": len(affinity_before)}
        line["library"] = _revelib.build_info()
        if world == 1 and not args.no_cpu_baseline:
            os.sched_setaffinity(0, affinity_before)      # the CPU baseline is the box's host cores, not the GPU's neighbours only
            line["cpu_baseline"] = cpu_baseline(weights, frames_np0, W, H)
        print(json.dumps(line), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


def C_cpulist(_revelib, device):
    """the CPUs next to a GPU as the library reads them from sysfs (reve_device_cpulist), "" if unknown"""
    import ctypes
    buf = ctypes.create_string_buffer(512)
    return buf.value.decode() if _revelib.load().reve_device_cpulist(device, buf, len(buf)) >= 0 else ""


if __name__ == "__main__":
    main()
