"""GPU: several small frames per launch (option "batch", on by default).

A 960x540 frame gives the pair kernel 16 strips x 16 segments of 34 rows, a 100x100 frame twelve units for 256 CUs.  Frames that
small are laid one below the other on ONE canvas (planes sharing their 1-pixel zero borders: the mechanism of tiled frames) and go
through the kernel chain together — reve_upscale_rgb8_device_batch, and the submit / wait ring by itself.  The bytes must be those
of one frame per launch, whatever the batch's size and however it was cut."""
import os

import numpy as np
import pytest
import torch

from oracle import ref
from reve_amd import synth
from reve_amd.upscaler import ReveError, Upscaler

pytestmark = pytest.mark.gpu
BUSY = -7


@pytest.fixture(scope="module")
def ups(model_bytes):
    made = {}

    def get(scale, batch):
        if (scale, batch) not in made:
            p, b = model_bytes(scale)
            up = Upscaler(scale, param=p, bin=b)
            up.set_option("batch", batch)
            made[(scale, batch)] = up
        return made[(scale, batch)]

    yield get
    for up in made.values():
        up.close()


def _device_batch(up, frames, scale):
    h, w, _ = frames[0].shape
    src = [torch.from_numpy(f).cuda() for f in frames]
    dst = [torch.empty((h * scale, w * scale, 3), dtype=torch.uint8, device="cuda") for _ in frames]
    up.upscale_device_batch([t.data_ptr() for t in src], [t.data_ptr() for t in dst], w, h)
    up.sync()
    return [t.cpu().numpy() for t in dst]


@pytest.mark.parametrize("w,h,scale", [(100, 100, 2), (256, 256, 2), (640, 480, 2), (37, 29, 3), (200, 131, 4), (63, 5, 2)])
def test_batches_write_the_bytes_of_single_launches(ups, weights, w, h, scale):
    one, many = ups(scale, 0), ups(scale, 1)
    frames = [synth.toon_frame(i, w, h) if i & 1 else synth.noise_frame(i, w, h) for i in range(21)]
    want = [one.upscale(f) for f in frames[:21]]
    assert one.get_option("batch_frames") == 1
    many.upscale(frames[0])
    k = many.get_option("batch_frames")
    assert 2 <= k <= 16
    for n in sorted({1, 2, k - 1, k, k + 1, 21}):          # a lone frame, partial batches, a full one, a full one and a rest
        got = _device_batch(many, frames[:n], scale)
        for i in range(n):
            assert np.array_equal(got[i], want[i]), (w, h, scale, n, i)
    d = np.abs(want[3].astype(np.int32) - ref.upscale(weights(scale), frames[3]).astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 0.01


def test_frames_that_fill_the_gpu_alone_are_not_batched(ups):
    up = ups(2, 1)
    up.upscale(synth.noise_frame(0, 1920, 1080))
    assert up.get_option("batch_frames") == 1
    up.upscale(synth.noise_frame(0, 960, 540))
    assert up.get_option("batch_frames") == 4


def test_the_ring_collects_batches_by_itself(ups, model_bytes):
    model_bytes_of = model_bytes
    one, many = ups(2, 0), ups(2, 1)
    w, h = 160, 90
    frames = [synth.toon_frame(i, w, h) for i in range(40)]
    want = [one.upscale(f) for f in frames]
    many.upscale(frames[0])
    k = many.get_option("batch_frames")
    outs = [np.empty((2 * h, 2 * w, 3), np.uint8) for _ in frames]
    # the ring takes two batches before it is full ...
    for i in range(2 * k):
        many.submit(i, frames[i], outs[i])
    with pytest.raises(ReveError) as e:
        many.submit(99, frames[0], np.empty_like(outs[0]))
    assert e.value.code == BUSY
    # ... frames come back in the order they went in, whatever batch they travelled in
    done = [many.wait() for _ in range(3)]
    assert done == [0, 1, 2]
    for i in range(2 * k, 40):                       # a steady state: retire one, submit one
        many.submit(i, frames[i], outs[i])
        done.append(many.wait())
    while len(done) < 40:                            # the tail: a partial batch that only reve_wait can release
        done.append(many.wait())
    assert done == list(range(40))
    for a, b in zip(outs, want):
        assert np.array_equal(a, b)
    # a lone frame is not held back for ever: reve_wait launches its batch of one
    many.submit(7, frames[7], outs[7])
    assert many.wait() == 7 and np.array_equal(outs[7], want[7])
    # an explicit ring depth is kept (and caps the batch): the fourth frame is refused, three travel together
    p, b = model_bytes_of(2)
    with Upscaler(2, param=p, bin=b, ring_depth=3) as three:
        for i in range(3):
            three.submit(i, frames[i], outs[i])
        with pytest.raises(ReveError) as e:
            three.submit(3, frames[3], outs[3])
        assert e.value.code == BUSY
        assert [three.wait() for _ in range(3)] == [0, 1, 2] and all(np.array_equal(outs[i], want[i]) for i in range(3))
    # another size with frames in flight: refused, as before
    many.submit(1, frames[1], outs[1])
    with pytest.raises(ReveError) as e:
        many.submit(2, synth.toon_frame(0, 64, 64), np.empty((128, 128, 3), np.uint8))
    assert e.value.code == BUSY
    assert many.wait() == 1


def test_directory_of_small_frames(ups, weights, tmp_path):
    from reve_amd.upscaler import png_read, png_write
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir()
    outd.mkdir()
    frames = [synth.toon_frame(i, 120, 80) for i in range(37)]
    for i, f in enumerate(frames):
        png_write(str(ind / f"frame{i + 1:08d}.png"), f)
    seen = []
    n = ups(2, 1).upscale_segment(str(ind), str(outd), on_done=lambda idx, a, b: seen.append(idx))
    assert n == 37 and seen == list(range(37))
    one = ups(2, 0)
    for i in (0, 17, 36):
        assert np.array_equal(png_read(str(outd / f"frame{i + 1:08d}.png")), one.upscale(frames[i]))


def test_a_ring_of_three_launches_partial_batches_and_counts_frames_when_they_retire(model_bytes):
    """A caller that keeps THREE frames in flight whatever the batch (the `reve` CLI's pipe lanes: reve_config.ring_depth = 3): the
    library launches a partial batch as soon as the GPU has nothing to do instead of holding frames for a batch this ring can never
    fill; every frame carries the bytes of the one-frame path; reve_stats.frames_done counts a frame when reve_wait
    returns it (round 5: it used to count at enqueue), device-path frames when reve_sync has seen the stream drain; and
    reve_upscale_rgb8_device answers REVE_E_BUSY while ring frames are in flight (a re-configure would pull the geometry from
    under them)."""
    from reve_amd.upscaler import pinned_array, free_pinned
    p, b = model_bytes(2)
    w, h, n = 256, 256, 40
    frames = [synth.noise_frame(i, w, h) if i % 3 else synth.toon_frame(i, w, h) for i in range(n)]
    with Upscaler(2, param=p, bin=b) as one:
        one.set_option("batch", 0)
        want = [one.upscale(f) for f in frames]
    with Upscaler(2, param=p, bin=b, ring_depth=3) as up:
        hin = [pinned_array((h, w, 3)) for _ in range(3)]
        hout = [pinned_array((2 * h, 2 * w, 3)) for _ in range(3)]
        done = 0
        for i in range(n):
            if i >= 3:
                assert up.wait() == i - 3
                done += 1
                assert np.array_equal(hout[(i - 3) % 3], want[i - 3]), i - 3
                assert up.stats()["frames_done"] == done
            hin[i % 3][...] = frames[i]
            up.submit(i, hin[i % 3], hout[i % 3])
            if i == 5:
                t = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
                o = torch.empty((2 * h, 2 * w, 3), dtype=torch.uint8, device="cuda")
                with pytest.raises(ReveError) as e:
                    up.upscale_device(t.data_ptr(), w, h, o.data_ptr())
                assert e.value.code == BUSY
        assert up.get_option("batch_frames") == 16          # the geometry still batches; the ring's depth caps what a launch holds
        for i in range(n - 3, n):
            assert up.wait() == i
            assert np.array_equal(hout[i % 3], want[i]), i
        assert up.stats()["frames_done"] == n
        # device path: counted when the stream is known to have drained
        src = torch.from_numpy(frames[0]).cuda()
        dst = torch.empty((2 * h, 2 * w, 3), dtype=torch.uint8, device="cuda")
        for _ in range(5):
            up.upscale_device(src.data_ptr(), w, h, dst.data_ptr())
        assert n <= up.stats()["frames_done"] <= n + 5
        up.sync()
        assert up.stats()["frames_done"] == n + 5 and np.array_equal(dst.cpu().numpy(), want[0])
        for a in hin + hout:
            free_pinned(a)


def test_a_failed_batch_launch_is_reported_per_frame_and_the_ring_recovers(model_bytes):
    """a launch that fails after its slots left the pending list used to leave their completion events unrecorded —
    reve_wait then returned at once, reported the frame finished and counted it, with `dst` never written.  The failure now stays
    on the batch's slots: reve_wait returns it once per frame (with the frame's id), counts nothing, and the ring is empty and
    usable afterwards.  The failure is injected between the chain and its events (option "debug_fail_launch")."""
    import ctypes as C
    p, b = model_bytes(2)
    w, h = 160, 90
    frames = [synth.toon_frame(i, w, h) for i in range(6)]
    with Upscaler(2, param=p, bin=b) as up:
        want = [up.upscale(f) for f in frames]
        k = up.get_option("batch_frames")
        assert k >= 4
        done0 = up.stats()["frames_done"]
        outs = [np.full((2 * h, 2 * w, 3), 0x5A, np.uint8) for _ in frames]
        up.set_option("debug_fail_launch", 1)
        for i in range(3):
            up.submit(i, frames[i], outs[i])          # (a partial batch: waits, uploaded)
        fid = C.c_uint64(99)
        codes = [up._lib.reve_wait(up._h, C.byref(fid)) for _ in range(1)]
        assert codes == [-4] and fid.value == 0 and b"was not upscaled" in up._lib.reve_last_error(up._h)      # REVE_E_HIP, frame 0
        for expect in (1, 2):
            assert up._lib.reve_wait(up._h, C.byref(fid)) == -4 and fid.value == expect
        assert up._lib.reve_wait(up._h, C.byref(fid)) == BUSY            # nothing in flight any more
        assert up.stats()["frames_done"] == done0 and all((o == 0x5A).all() for o in outs[:3])      # nothing counted, nothing written
        up._inflight.clear()
        for i in range(6):                                               # the context is usable: the next batches are right
            up.submit(i, frames[i], outs[i])
        assert [up.wait() for _ in range(6)] == list(range(6))
        assert all(np.array_equal(outs[i], want[i]) for i in range(6)) and up.stats()["frames_done"] == done0 + 6
