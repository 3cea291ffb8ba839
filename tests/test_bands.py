"""Band split of one descent (include/tpose_hip.h: tp_band_attach): several contexts run the patches
of ONE plan between them and exchange vertex positions through each other's mailboxes.  On the one-GPU test box the bands are
two contexts (two streams) of one process on device 0, 100 patches each -- the arithmetic of the seam and the hand-over
protocol are the same as between two GPUs; what the box cannot show is the latency of the link."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from tpose_amd import capi, synth  # noqa: E402
from util import RATE, case  # noqa: E402

pytestmark = pytest.mark.gpu


class _Box:
    """a band's mailbox: fine-grained device memory from tp_band_mailbox_alloc (what bands on different GPUs need), freed with its context"""

    def __init__(self, ctx, nbytes):
        self.ctx, self.ptr = ctx, ctx.band_mailbox_alloc(nbytes)

    def data_ptr(self):
        return self.ptr


def banded_contexts(W, H, img, imgB, pts, tris, colors, n_bands, patches):
    cap_p, cap_t = pts.shape[0] + 64, tris.shape[0] + 64
    nbytes = capi.band_mailbox_bytes(cap_p, cap_t)
    ctxs = [capi.Context(0, W, H) for _ in range(n_bands)]
    boxes = [_Box(ctx, nbytes) for ctx in ctxs]
    for b, ctx in enumerate(ctxs):
        assert ctx.info(capi.INFO_BOX_FINEGRAINED) == 1
        ctx.set_image(capi.IMAGE_A, img)
        ctx.set_image(capi.IMAGE_B, imgB)
        ctx.upload(pts, tris, colors)
        ctx.band_attach(b, n_bands, [bx.data_ptr() for bx in boxes], nbytes, cap_p, cap_t, patches)
        # (the census of resident workgroups wants the device to itself: on a shared device, before any band spins in a launch)
        ctx.prepare(capi.default_params(1 if colors is not None else 0))
    return ctxs, boxes


@pytest.mark.parametrize("flavour", [0, 1])
@pytest.mark.parametrize("n_bands,patches", [(2, 8), (3, 5)])
def test_bands_descend_like_one_context(flavour, n_bands, patches):
    """2 and 3 bands, calls of several lengths (single and chunked launches, odd numbers): every band ends with the
    oracle's positions, bit for bit; the emitted buffers of a band are the oracle's on the entries of its own patches"""
    W, H = 300, 200
    img, imgB, pts, tris, ratio, colors = case(W, H, (15, 5))
    ctxs, boxes = banded_contexts(W, H, img, imgB, pts, tris, colors if flavour else None, n_bands, patches)
    p = capi.default_params(flavour)
    total = 0
    for n in (6, 131, 700, 5):
        for ctx in ctxs:
            ctx.iterate(p, n)       # (enqueued: the bands' launches run side by side)
        for ctx in ctxs:
            ctx.synchronize()
        total += n
        ref = O.iterate(imgB if flavour else img, pts, tris, flavour, ratio, RATE[flavour], total, colors=colors if flavour else None, literal=False)
        covered = np.zeros(ref["ten"].shape[0], bool)
        for ctx in ctxs:
            assert ctx.info(9) == 0, "a band gave up"
            assert ctx.info(capi.INFO_PERSIST_ITERS) == total and ctx.info(capi.INFO_PATCHES) == n_bands * patches
            assert np.array_equal(ctx.retrieve(capi.BUF_POINTS).view(np.uint32), ref["points"].view(np.uint32))
            ten = ctx.retrieve(capi.BUF_TENERGY)
            covered |= ten == ref["ten"]
        assert covered.all()        # every variant's energy was written by the band that owns it
    for ctx in ctxs:
        ctx.close()


@pytest.mark.parametrize("flavour,threshold,cap", [(0, 1e-4, 300), (1, 1e-6, 90)])
def test_bands_run_the_reference_loop_to_its_convergence_test(flavour, threshold, cap):
    """tp_iterate_until on two bands (one host thread each: the call blocks on the other band's frames): the same number of
    frames, the same running total and -- the last frame is run whole on every band -- the same buffers as the unsplit call"""
    import threading
    W, H, grid = 300, 200, (15, 5)
    img = synth.photo_contrast(synth.voronoi_raster(W, H, seed=7, sites=12), 0.3)
    imgB = synth.displaced_raster(img, amp=6.0)
    ratio = float(np.float32(W) / np.float32(H))
    pts, tris, _ = synth.grid_triangulation(grid[0], grid[1], ratio=ratio)
    colors = synth.mean_colors(img, pts, tris, ratio) if flavour else None
    p = capi.default_params(flavour)
    one = capi.Context(0, W, H)
    one.set_image(capi.IMAGE_A, img)
    one.set_image(capi.IMAGE_B, imgB)
    one.upload(pts, tris, colors)
    want = [one.iterate_until(p, cap, threshold, 1.0)]
    want.append(one.iterate_until(p, cap, threshold, want[0][1]))   # (a second leg: the running total carries over)
    ctxs, boxes = banded_contexts(W, H, img, imgB, pts, tris, colors, 2, 8)
    got = [[None, None], [None, None]]

    def leg(b):
        got[b][0] = ctxs[b].iterate_until(p, cap, threshold, 1.0)
        got[b][1] = ctxs[b].iterate_until(p, cap, threshold, got[b][0][1])

    th = [threading.Thread(target=leg, args=(b,)) for b in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    bits = lambda r: (r[0], np.float32(r[1]).view(np.uint32), np.float32(r[2]).view(np.uint32))  # noqa: E731
    for b in range(2):
        assert [bits(r) for r in got[b]] == [bits(r) for r in want], (got[b], want)
        assert ctxs[b].info(9) == 0 and ctxs[b].info(capi.INFO_PERSIST_ITERS) > 0
        for buf in (capi.BUF_POINTS, capi.BUF_TENERGY, capi.BUF_COLNUM, capi.BUF_GRADIENT):
            assert np.array_equal(ctxs[b].retrieve(buf).view(np.uint32), one.retrieve(buf).view(np.uint32)), buf
    for c in ctxs + [one]:
        c.close()


def test_band_that_never_shows_up_is_survived():
    """band 1 never iterates: band 0 waits a second in its launch, gives up without having changed anything, and runs the
    call on its own -- the oracle's bits again, also when a short call was enqueued right behind the launch that gave up"""
    W, H = 300, 200
    img, imgB, pts, tris, ratio, colors = case(W, H, (15, 5))
    ctxs, boxes = banded_contexts(W, H, img, imgB, pts, tris, None, 2, 8)
    p = capi.default_params(0)
    ctxs[0].iterate(p, 40)
    ctxs[0].iterate(p, 2)      # (too short for a persistent launch: it must not overtake the 40 that are about to be run again)
    ctxs[0].synchronize()
    ref = O.iterate(img, pts, tris, 0, ratio, RATE[0], 42, literal=False)
    assert ctxs[0].info(9) == 1
    assert np.array_equal(ctxs[0].retrieve(capi.BUF_POINTS).view(np.uint32), ref["points"].view(np.uint32))
    assert np.array_equal(ctxs[0].retrieve(capi.BUF_TENERGY), ref["ten"])
    for ctx in ctxs:
        ctx.close()


def test_bands_on_two_devices():
    """the same on TWO GPUs (skipped on a one-GPU box): band 0 on device 0, band 1 on device 1, each with its own fine-grained mailbox
    (tp_band_mailbox_alloc) that the other device's kernel writes while this one polls it; tp_band_attach enables peer access.  The
    oracle's bits, no launch given up (a coarse-grained mailbox would wait out its second in every launch and count as given up)."""
    if capi.device_count() < 2:
        pytest.skip("needs two GPUs")
    W, H = 300, 200
    img, imgB, pts, tris, ratio, colors = case(W, H, (15, 5))
    cap_p, cap_t = pts.shape[0] + 64, tris.shape[0] + 64
    nbytes = capi.band_mailbox_bytes(cap_p, cap_t)
    ctxs = [capi.Context(d, W, H) for d in (0, 1)]
    boxes = [ctx.band_mailbox_alloc(nbytes) for ctx in ctxs]
    for b, ctx in enumerate(ctxs):
        ctx.set_image(capi.IMAGE_A, img)
        ctx.upload(pts, tris, None)
        ctx.band_attach(b, 2, boxes, nbytes, cap_p, cap_t, 8)
        ctx.prepare(capi.default_params(0))
    p = capi.default_params(0)
    total = 0
    for n in (6, 131, 40):
        for ctx in ctxs:
            ctx.iterate(p, n)
        for ctx in ctxs:
            ctx.synchronize()
        total += n
        ref = O.iterate(img, pts, tris, 0, ratio, RATE[0], total, literal=False)
        for ctx in ctxs:
            assert ctx.info(capi.INFO_PERSIST_FAILURES) == 0, "a band gave up: the mailboxes are not visible across the devices"
            assert np.array_equal(ctx.retrieve(capi.BUF_POINTS).view(np.uint32), ref["points"].view(np.uint32))
    for ctx in ctxs:
        ctx.close()
