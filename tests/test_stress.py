"""One context under a random interleaving of everything the host side of the persistent path does -- short and long tp_iterate calls
(persistent launches, chunks, snapshots, re-plans on the calling thread and on the worker thread), single frames (two-kernel path, frame
mirror, lazily filed edge endpoints), read-backs in between, uploads of the moved mesh, tp_iterate_until, a change of image and of dp --
with the oracle following every step: after each operation that reads something back the bits must agree.

What it is after: the status-word replay logic, the pinned snapshots, the re-plan worker and the lazily filed `epos` are the parts of
tp_persist_host.hip / tp_replan.hip most likely to hide an ordering bug, and no single-purpose test walks through them in this order."""
import numpy as np
import pytest

from oracle import oracle as O
from tpose_amd import capi, synth
from util import RATE

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("flavour,seed", [(0, 1), (1, 2), (0, 3)])
def test_interleaved_calls_follow_the_oracle(flavour, seed):
    rng = np.random.default_rng(seed)
    W, H, grid = 320, 240, (16, 8)
    img = synth.photo_contrast(synth.voronoi_raster(W, H, seed=40 + seed, sites=14), 0.35)
    img2 = synth.photo_contrast(synth.voronoi_raster(W, H, seed=90 + seed, sites=9), 0.35)
    ratio = float(np.float32(W) / np.float32(H))
    pts, tris, _ = synth.grid_triangulation(grid[0], grid[1], ratio=ratio)
    colors = synth.mean_colors(img, pts, tris, ratio) if flavour else None
    slot = capi.IMAGE_B if flavour else capi.IMAGE_A
    ctx = capi.Context(0, W, H)
    ctx.set_image(capi.IMAGE_A, img)
    ctx.set_image(capi.IMAGE_B, img)
    ctx.upload(pts, tris, colors)
    p = capi.default_params(flavour)
    # (a step three times the reference's: vertices drift by pixels within a few hundred grad-iters, so plans are cut again)
    p.rate = RATE[flavour] * 3.0
    cur_img, cur_pts, dp = img, pts.copy(), None
    last = None   # the oracle's buffers after the last grad-iter

    def advance(n):
        nonlocal cur_pts, last
        last = O.iterate(cur_img, cur_pts, tris, flavour, ratio, p.rate, n, colors=colors, dp_=dp, literal=False)
        cur_pts = last["points"]

    def check(tag, buffers=True):
        got = ctx.retrieve(capi.BUF_POINTS)
        assert np.array_equal(got.view(np.uint32), cur_pts.view(np.uint32)), tag
        if buffers and last is not None:
            assert np.array_equal(ctx.retrieve(capi.BUF_TENERGY), last["ten"]), tag
            assert np.array_equal(ctx.retrieve(capi.BUF_GRADIENT), last["gr"]), tag

    ops = ["short", "long", "frame", "frames3", "readback", "upload", "until", "image", "dp", "piecewise", "long"]
    done = 0
    for step in range(22):
        op = ops[int(rng.integers(len(ops)))] if step > 2 else ["long", "short", "frame"][step]
        if op == "short":
            n = int(rng.integers(4, 40))
            ctx.iterate(p, n); advance(n); done += n
            if rng.random() < 0.5:
                check("short %d" % n)
        elif op == "long":
            n = int(rng.integers(260, 420))
            ctx.iterate(p, n); advance(n); done += n
            if rng.random() < 0.5:
                check("long %d" % n)
        elif op == "frame":
            ctx.iterate(p, 1); advance(1); done += 1
            check("single frame")
        elif op == "frames3":
            ctx.iterate(p, 3); advance(3); done += 3
        elif op == "readback":
            check("read-back")
        elif op == "upload":
            moved = ctx.retrieve(capi.BUF_POINTS)
            assert np.array_equal(moved.view(np.uint32), cur_pts.view(np.uint32)), "before upload"
            ctx.upload(moved, tris, colors)
            last = None
        elif op == "until":
            cap = int(rng.integers(5, 60))
            n, tot, rel = ctx.iterate_until(p, cap, 0.0, 1.0)
            assert n == cap
            advance(cap); done += cap
            check("iterate_until %d" % cap)
        elif op == "image":
            cur_img = img2 if cur_img is img else img
            ctx.set_image(slot, cur_img)
        elif op == "dp":
            dp = None if dp is not None else 0.02
            p.dp = 0.0 if dp is None else dp
        elif op == "piecewise":
            ctx.set_dp(0.0 if dp is None else dp)
            ctx.accumulate(flavour, slot); ctx.energy(flavour); ctx.shift(p.rate)
            advance(1); done += 1
            check("piecewise frame", buffers=False)
    check("end")
    print("stress: %d grad-iters, %d inside persistent launches, %d plans cut again" % (done, ctx.info(capi.INFO_PERSIST_ITERS), ctx.info(capi.INFO_REPLANS)))
    assert ctx.info(capi.INFO_PERSIST_ITERS) > 0 and ctx.info(capi.INFO_PERSIST_FAILURES) == 0
    ctx.close()
