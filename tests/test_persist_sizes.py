"""GPU parity of the PERSISTENT kernel (k_persist: what bench.py times) against the oracle AT THE BENCHMARKED SIZES.

Every case runs tp_iterate of four or more grad-iters with persistent launches on, asserts that the grad-iters really ran
inside persistent launches (INFO_PERSIST_ITERS), and compares `tenergy`, `colnum`, `colacc`, `gradient` and the float32
vertex positions with oracle.iterate(..., literal=False) bit for bit (tolerance 0 ulp):
  * 2048^2 / 3000 triangles on bench.py's raster (contrast 0.1) and on SURVEY section 8(d)'s full-contrast raster, both flavours;
  * the same through tp_iterate_until (the instantiation that forms all 13 variants in every grad-iter);
  * 4096^2 / 12 000 (BASELINE config 4's element): more lane-items per workgroup than threads keep records for;
  * 674 x 449 / 150 triangles x 200 grad-iters (BASELINE config 1 as written: software/triangulate/main.cpp:53,190-204);
  * hostile vertex sets at the metric size with enough grad-iters for the persistent path.
"""
import numpy as np
import pytest

from oracle import oracle as O
from tpose_amd import capi, synth
from util import RATE

pytestmark = pytest.mark.gpu


def _setup(W, H, NT, flavour, contrast):
    img, pts, tris, he, ratio = synth.workload(W, H, NT, contrast=contrast)
    imgB, colors = None, None
    if flavour == 1:
        imgB = synth.displaced_raster(img)
        colors = synth.mean_colors(img, pts, tris, ratio)
    ctx = capi.Context(0, W, H)
    ctx.set_image(capi.IMAGE_A, img)
    if flavour == 1:
        ctx.set_image(capi.IMAGE_B, imgB)
    ctx.upload(pts, tris, colors)
    return ctx, (imgB if flavour == 1 else img), pts, tris, ratio, colors


def _compare(ctx, ref, flavour, tag=""):
    assert np.array_equal(ctx.retrieve(capi.BUF_TENERGY), ref["ten"]), "tenergy " + tag
    assert np.array_equal(ctx.retrieve(capi.BUF_COLNUM), ref["cn"]), "colnum " + tag
    if flavour == 0:
        assert np.array_equal(ctx.retrieve(capi.BUF_COLACC)[:, :3], ref["ca"][:, :3]), "colacc " + tag
    assert np.array_equal(ctx.retrieve(capi.BUF_GRADIENT), ref["gr"]), "gradient " + tag
    assert np.array_equal(ctx.retrieve(capi.BUF_POINTS).view(np.uint32), ref["points"].view(np.uint32)), "points " + tag


@pytest.mark.parametrize("contrast", [0.1, 1.0])
@pytest.mark.parametrize("flavour", [0, 1])
def test_persistent_is_the_oracle_at_the_metric_size(flavour, contrast):
    """2048^2 / 3000: 12 grad-iters in ONE persistent launch (k_persist<RR, 0>, 256 patches -- the kernel and size BENCH times)"""
    iters = 12
    ctx, sweep, pts, tris, ratio, colors = _setup(2048, 2048, 3000, flavour, contrast)
    p = capi.default_params(flavour)
    ctx.prepare(p)
    ctx.iterate(p, iters)
    ctx.synchronize()
    assert ctx.info(capi.INFO_PERSIST_ITERS) == iters and ctx.info(capi.INFO_PATCHES) >= 128
    assert ctx.info(9) == 0   # no launch gave up (nothing was replayed on the two-kernel path)
    ref = O.iterate(sweep, pts, tris, flavour, ratio, RATE[flavour], iters, colors=colors, literal=False)
    _compare(ctx, ref, flavour)
    # ... and a second call continues from there (another launch, the plan and the mailbox epochs carried over)
    ctx.iterate(p, 5)
    ref2 = O.iterate(sweep, ref["points"], tris, flavour, ratio, RATE[flavour], 5, colors=colors, literal=False)
    _compare(ctx, ref2, flavour, "second call")
    assert ctx.info(capi.INFO_PERSIST_ITERS) == iters + 5
    ctx.close()


@pytest.mark.parametrize("flavour", [0, 1])
def test_iterate_until_at_the_metric_size(flavour):
    """2048^2 / 3000 through tp_iterate_until (k_persist<RR, 1>: base lines, hence all 13 variants, in EVERY grad-iter): a
    threshold nothing meets, so exactly `frames` frames run; the state is the oracle's after as many grad-iters"""
    frames = 9
    ctx, sweep, pts, tris, ratio, colors = _setup(2048, 2048, 3000, flavour, 0.1)
    p = capi.default_params(flavour)
    n, tot, rel = ctx.iterate_until(p, frames, 0.0, 1.0)
    assert n == frames
    assert ctx.info(capi.INFO_PERSIST_ITERS) >= frames - 1   # (the last frame runs once more on the two-kernel path)
    ref = O.iterate(sweep, pts, tris, flavour, ratio, RATE[flavour], frames, colors=colors, literal=False)
    _compare(ctx, ref, flavour)
    # the running total is geterr's float32 sum of the LAST frame's base energies (source/triangulation.hpp:653-674)
    newerr = np.float32(0)
    for v in ref["ten"][: tris.shape[0]].astype(np.float32):
        newerr = np.float32(newerr + v)
    assert np.float32(tot).view(np.uint32) == newerr.view(np.uint32)
    ctx.close()


@pytest.mark.parametrize("flavour,contrast", [(0, 0.1), (1, 0.1), (0, 1.0)])
def test_persistent_is_the_oracle_at_the_batch_size(flavour, contrast):
    """4096^2 / 12 000 (BASELINE config 4's element): ~3400 lane-items per workgroup, more than its threads keep records
    for -- the overflow path of the walk; on bench.py's raster and on SURVEY 8(d)'s raster as written"""
    iters = 4
    ctx, sweep, pts, tris, ratio, colors = _setup(4096, 4096, 12000, flavour, contrast)
    p = capi.default_params(flavour)
    ctx.iterate(p, iters)
    ctx.synchronize()
    assert ctx.info(capi.INFO_PERSIST_ITERS) == iters and ctx.info(9) == 0
    assert ctx.info(capi.INFO_PLAN_ROWS) > 16   # (... and the rows of a lane beyond its registers, kept in LDS)
    ref = O.iterate(sweep, pts, tris, flavour, ratio, RATE[flavour], iters, colors=colors, literal=False)
    _compare(ctx, ref, flavour)
    ctx.close()


@pytest.mark.parametrize("size,contrast", [(3072, 0.1), (3072, 1.0), (2300, 0.1), (2700, 0.1), (3500, 1.0)])
def test_rows_beyond_the_registers_are_the_oracle(size, contrast):
    """3072^2 / 3000: edge lines half again as long as at the metric size -- a plan of more than 16 rows per lane: rows 16..23 of every thread's
    lane-item keep their records in LDS (pk_walk_lds_rows; k_persist<24, 0>, 23 rows per lane: both words of a slot's crossing columns), and
    at 3500^2 (24 rows per lane) some lane-items have no slot at all.  14 grad-iters in one launch, then 6 in a second, warm one: records
    kept, dropped and fetched again under the oracle.  2300^2: 17-18 rows per lane, the instantiation with two rows in LDS (k_persist<18, 0>);
    2700^2: 20 rows per lane in the 24-row instantiation (its waves skip the second four LDS rows)."""
    ctx, sweep, pts, tris, ratio, colors = _setup(size, size, 3000, 0, contrast)
    p = capi.default_params(0)
    ctx.iterate(p, 14)
    ctx.synchronize()
    assert ctx.info(capi.INFO_PERSIST_ITERS) == 14 and ctx.info(9) == 0
    rows = ctx.info(capi.INFO_PLAN_ROWS)
    assert {2300: rows in (17, 18), 2700: rows in (19, 20), 3072: rows > 20, 3500: rows == 24}[size], rows
    ref = O.iterate(sweep, pts, tris, 0, ratio, RATE[0], 14, colors=colors, literal=False)
    _compare(ctx, ref, 0)
    ctx.iterate(p, 6)
    ref2 = O.iterate(sweep, ref["points"], tris, 0, ratio, RATE[0], 6, colors=colors, literal=False)
    _compare(ctx, ref2, 0, "second call")
    assert ctx.info(capi.INFO_PERSIST_ITERS) == 20 and ctx.info(9) == 0
    ctx.close()


@pytest.mark.parametrize("contrast", [0.1, 1.0])
def test_config1_as_written(contrast):
    """BASELINE config 1: the 674 x 449 window of fruit.png (1011 x 674 / 1.5), 150 triangles, 200 grad-iters, triangulate
    flavour (software/triangulate/main.cpp:53, 190-204) -- one call, and the same in calls of 37 grad-iters"""
    W, H, NT, iters = 674, 449, 150, 200
    ctx, sweep, pts, tris, ratio, colors = _setup(W, H, NT, 0, contrast)
    p = capi.default_params(0)
    ctx.iterate(p, iters)
    ctx.synchronize()
    assert ctx.info(capi.INFO_PERSIST_ITERS) == iters and ctx.info(9) == 0
    ref = O.iterate(sweep, pts, tris, 0, ratio, RATE[0], iters, literal=False)
    _compare(ctx, ref, 0)
    ctx.upload(pts, tris, None)
    done = 0
    while done < iters:
        k = min(37, iters - done)
        ctx.iterate(p, k)
        done += k
    _compare(ctx, ref, 0, "calls of 37")
    ctx.close()


@pytest.mark.parametrize("kind", ["huge", "allsame", "concentrated", "nan", "inf"])
def test_hostile_vertex_sets_on_the_persistent_path(kind):
    """the hostile vertex sets of test_hip_parity.py with enough grad-iters (6) for the persistent path: no fault, and --
    where the positions are numbers -- the oracle's bits.  A plan may refuse such a mesh (then the call runs on the
    two-kernel path); when it does not, the grad-iters must have run inside persistent launches."""
    W = H = 2048
    img, pts, tris, he, ratio = synth.workload(W, H, 3000)
    bad = pts.copy()
    sel = (np.arange(bad.shape[0]) % 7 == 5)
    if kind == "nan":
        bad[sel] = np.nan
    elif kind == "inf":
        bad[sel] = np.inf
    elif kind == "huge":
        bad *= np.float32(1e6)
    elif kind == "allsame":
        bad[:] = 0
    else:
        bad[4:] = bad[4:] * np.float32(0.1) * np.array([1.0, 0.5], np.float32) + np.float32(0.3)
    iters = 6
    ctx = capi.Context(0, W, H)
    ctx.set_image(capi.IMAGE_A, img)
    ctx.upload(bad, tris, None)
    p = capi.default_params(capi.TRIANGULATE)
    ctx.iterate(p, iters)
    ctx.synchronize()
    if ctx.info(capi.INFO_PATCHES) > 0:
        assert ctx.info(capi.INFO_PERSIST_ITERS) == iters
    if kind in ("huge", "allsame", "concentrated"):
        ref = O.iterate(img, bad, tris, O.TRIANGULATE, ratio, RATE[0], iters, literal=False)
        _compare(ctx, ref, 0, kind)
    else:
        assert ctx.retrieve(capi.BUF_POINTS).shape == bad.shape
    ctx.close()


@pytest.mark.parametrize("flavour", [0, 1])
def test_a_launch_that_gives_up_is_run_again(flavour):
    """A plain persistent launch finishes itself: it writes the positions it ends with into the context's second position buffer and the
    host swaps the two.  TP_OPT_INJECT_GIVE_UP makes one workgroup of a launch give up before its LAST grad-iter -- every other workgroup
    has written its vertices by then -- with another call already enqueued behind it: both are run again on the two-kernel path from the
    positions the launch started with, the oracle's bits."""
    ctx, sweep, pts, tris, ratio, colors = _setup(2048, 2048, 3000, flavour, 0.1)
    p = capi.default_params(flavour)
    ctx.iterate(p, 8)
    ctx.iterate(p, 6)
    ctx.set_option(capi.OPT_INJECT_GIVE_UP, 1)
    ctx.iterate(p, 7)    # gives up
    ctx.iterate(p, 5)    # behind it: does nothing
    ctx.synchronize()
    got26 = {b: ctx.retrieve(b) for b in (capi.BUF_TENERGY, capi.BUF_COLNUM, capi.BUF_COLACC, capi.BUF_GRADIENT, capi.BUF_POINTS)}
    assert ctx.info(capi.INFO_PERSIST_FAILURES) == 1 and ctx.info(capi.INFO_PERSIST_LAUNCHES) == 4
    # ... and the context carries on, two kernels per grad-iter for now (right away: the oracle is consulted afterwards, it takes its time)
    ctx.iterate(p, 9)
    ctx.synchronize()
    assert ctx.info(capi.INFO_PERSIST_LAUNCHES) == 4
    ref = O.iterate(sweep, pts, tris, flavour, ratio, RATE[flavour], 26, colors=colors, literal=False)
    assert np.array_equal(got26[capi.BUF_TENERGY], ref["ten"]) and np.array_equal(got26[capi.BUF_COLNUM], ref["cn"])
    assert np.array_equal(got26[capi.BUF_GRADIENT], ref["gr"]) and np.array_equal(got26[capi.BUF_POINTS].view(np.uint32), ref["points"].view(np.uint32))
    if flavour == 0:
        assert np.array_equal(got26[capi.BUF_COLACC][:, :3], ref["ca"][:, :3])
    ref2 = O.iterate(sweep, ref["points"], tris, flavour, ratio, RATE[flavour], 9, colors=colors, literal=False)
    _compare(ctx, ref2, flavour, "after the replay")
    # ... for a while: a collision is a transient thing, so 0.2 s after the first give-up persistent launches are tried again
    import time
    time.sleep(0.3)
    ctx.iterate(p, 10)
    ref3 = O.iterate(sweep, ref2["points"], tris, flavour, ratio, RATE[flavour], 10, colors=colors, literal=False)
    _compare(ctx, ref3, flavour, "persistent again")
    assert ctx.info(capi.INFO_PERSIST_LAUNCHES) == 5 and ctx.info(capi.INFO_PERSIST_FAILURES) == 1
    ctx.close()


def test_vertices_no_triangle_uses_keep_the_finishing_kernel():
    """shift.cs clamps every vertex i >= 4, used or not: with such vertices in the upload the persistent launches keep their small
    finishing kernel (which clamps them) instead of finishing themselves"""
    W, H, NT, iters = 674, 449, 150, 40
    img, pts, tris, he, ratio = synth.workload(W, H, NT, contrast=0.1)
    pts2 = np.concatenate([pts, np.array([[3.0, 0.5], [-0.2, -7.0], [0.3, 0.3]], np.float32)])
    n = pts.shape[0]
    ctx = capi.Context(0, W, H)
    ctx.set_image(capi.IMAGE_A, img)
    ctx.upload(pts2, tris, None)
    p = capi.default_params(0)
    ctx.iterate(p, iters)
    ctx.iterate(p, 7)
    assert ctx.info(capi.INFO_PERSIST_ITERS) == iters + 7
    ref = O.iterate(img, pts2, tris, 0, ratio, RATE[0], iters + 7, literal=False)
    _compare(ctx, ref, 0)
    got = ctx.retrieve(capi.BUF_POINTS)
    assert got[n, 0] == np.float32(ratio) and got[n + 1, 1] == np.float32(-1.0) and np.array_equal(got[n + 2], pts2[n + 2])
    ctx.close()


def test_two_contexts_on_one_device_take_turns():
    """Two contexts of one process with persistent launches of both pending at once (the two directions of a warp, a batch): each launch
    wants every compute unit, so they take turns on the device (an event behind each, a wait in front of the next context's) instead of
    being handed out half a grid each and waiting out their time limit.  No launch gives up, and both follow the oracle."""
    W = H = 2048
    ctxs, refs = [], []
    for seed in (11, 12):
        img, pts, tris, he, ratio = synth.workload(W, H, 3000, seed=seed, contrast=0.1)
        c = capi.Context(0, W, H)
        c.set_image(capi.IMAGE_A, img)
        c.upload(pts, tris, None)
        ctxs.append(c)
        refs.append((img, pts, tris, ratio))
    p = capi.default_params(0)
    for c in ctxs:
        c.prepare(p)
    total = 0
    for n in (9, 6, 11, 5, 8, 7):          # enqueued alternately, nobody waits in between
        for c in ctxs:
            c.iterate(p, n)
        total += n
    for c, (img, pts, tris, ratio) in zip(ctxs, refs):
        ref = O.iterate(img, pts, tris, 0, ratio, RATE[0], total, literal=False)
        _compare(c, ref, 0)
        assert c.info(capi.INFO_PERSIST_FAILURES) == 0 and c.info(capi.INFO_PERSIST_ITERS) == total
        c.close()


@pytest.mark.parametrize("first", [1, 2])
def test_a_small_mesh_whose_early_launch_gives_up(first):
    """tools/run_batch.py in the small (512^2, 150 triangles, warp flavour, two contexts of one process): the FIRST or second persistent
    launch of a context gives up; the calls are run again and the energies and positions are the oracle's -- also for a read-back right
    behind the call that gave up (round 4: such a read-back returned what the buffers held BEFORE the replay; seen once as a two-rank batch
    whose `energy_before` was 0)"""
    W = H = 512
    img, pts, tris, he, ratio = synth.workload(W, H, 150, seed=4001)
    imgB = synth.displaced_raster(img)
    colors = synth.mean_colors(img, pts, tris, ratio)
    ctxs = []
    for k in range(2):
        c = capi.Context(0, W, H)
        c.set_image(capi.IMAGE_A, img); c.set_image(capi.IMAGE_B, imgB)
        c.upload(pts, tris, colors)
        ctxs.append(c)
    p = capi.default_params(capi.WARP)
    ctxs[0].set_option(capi.OPT_INJECT_GIVE_UP, first)
    for c in ctxs:
        c.iterate(p, 16)
    # (read back right behind the call, nothing waited for in between: the read-back itself finds the launch that gave up, has its call run
    # again and must then return what THAT left -- tools/run_batch.py's `energy_before`)
    e0 = [c.retrieve(capi.BUF_TENERGY)[:150].astype(np.int64).sum() for c in ctxs]
    ref16 = O.iterate(imgB, pts, tris, 1, ratio, RATE[1], 16, colors=colors, literal=False)
    assert all(e == ref16["ten"][:150].astype(np.int64).sum() for e in e0), e0
    for c in ctxs:
        c.iterate(p, 48)
    for c in ctxs:
        c.synchronize()
    ref = O.iterate(imgB, pts, tris, 1, ratio, RATE[1], 64, colors=colors, literal=False)
    for k, c in enumerate(ctxs):
        _compare(c, ref, 1, "context %d" % k)
        assert c.retrieve(capi.BUF_TENERGY)[:150].astype(np.int64).sum() < e0[k]
    assert ctxs[0].info(capi.INFO_PERSIST_FAILURES) == 1 and ctxs[1].info(capi.INFO_PERSIST_FAILURES) == 0
    for c in ctxs:
        c.close()


def test_warm_launches_start_from_the_last_cut_and_never_after_an_upload_image_or_dp():
    """round 5: a persistent launch leaves the cut of its patches' lines and its threads' lane-items for the next launch on the same plan
    (tp_persist.hip, "carry"), which then neither counts chunks nor searches in its first grad-iter; its first grad-iter also fetches every
    record from the tiled copy of the table.  Same bits across warm launches (calls of 5, 7, 60 and 9 grad-iters: the lines are cut again
    INSIDE the third, 64 grad-iters after the first cut), and a launch behind tp_upload, tp_set_image or tp_set_dp is never warm."""
    W, H, NT = 640, 480, 3000
    img, pts, tris, he, ratio = synth.workload(W, H, NT, contrast=0.3)
    ctx = capi.Context(0, W, H)
    ctx.set_image(capi.IMAGE_A, img)
    ctx.upload(pts, tris, None)
    p = capi.default_params(0)
    ref_pts, warm = pts, 0
    for k, n in enumerate((5, 7, 60, 9)):
        ctx.iterate(p, n)
        ref = O.iterate(img, ref_pts, tris, 0, ratio, RATE[0], n, literal=False)
        _compare(ctx, ref, 0, "call %d" % k)
        ref_pts = ref["points"]
        assert ctx.info(capi.INFO_WARM_LAUNCHES) == warm + (1 if k > 0 else 0), k
        warm = ctx.info(capi.INFO_WARM_LAUNCHES)
    assert ctx.info(capi.INFO_PERSIST_FAILURES) == 0 and ctx.info(capi.INFO_PERSIST_ITERS) == 81
    # a new image: the next launch cuts for itself, the one behind it is warm again
    img2 = np.ascontiguousarray(img[::-1])
    ctx.set_image(capi.IMAGE_A, img2)
    ctx.iterate(p, 6)
    assert ctx.info(capi.INFO_WARM_LAUNCHES) == warm
    ref = O.iterate(img2, ref_pts, tris, 0, ratio, RATE[0], 6, literal=False)
    _compare(ctx, ref, 0, "after tp_set_image"); ref_pts = ref["points"]
    ctx.iterate(p, 6)
    assert ctx.info(capi.INFO_WARM_LAUNCHES) == warm + 1
    warm += 1
    ref = O.iterate(img2, ref_pts, tris, 0, ratio, RATE[0], 6, literal=False)
    _compare(ctx, ref, 0, "warm on the new image"); ref_pts = ref["points"]
    # tp_set_dp
    ctx.set_dp(0.02)
    ctx.iterate(p, 5)
    assert ctx.info(capi.INFO_WARM_LAUNCHES) == warm
    ref = O.iterate(img2, ref_pts, tris, 0, ratio, RATE[0], 5, literal=False)
    _compare(ctx, ref, 0, "after tp_set_dp"); ref_pts = ref["points"]
    # tp_upload of the moved mesh
    ctx.upload(ref_pts, tris, None)
    ctx.iterate(p, 5)
    assert ctx.info(capi.INFO_WARM_LAUNCHES) == warm
    ref = O.iterate(img2, ref_pts, tris, 0, ratio, RATE[0], 5, literal=False)
    _compare(ctx, ref, 0, "after tp_upload")
    ctx.close()


def test_an_ageing_descent_under_the_oracle():
    """Ageing in the driver-run suite: 640 x 480 / 3000 triangles at three times the reference's rate, 2100 grad-iters in calls of mixed
    lengths -- the lines are cut again every 64 grad-iters (32 times), launches hand their cuts to each other (warm launches), vertices drift
    far enough for the host to cut new plans -- and after EVERY call all four buffers and the float32 positions are the oracle's, 0 ulp."""
    W, H, NT = 640, 480, 3000
    img, pts, tris, he, ratio = synth.workload(W, H, NT, contrast=0.3)
    ctx = capi.Context(0, W, H)
    ctx.set_image(capi.IMAGE_A, img)
    ctx.upload(pts, tris, None)
    rate = np.float32(3.0) * np.float32(RATE[0])
    p = capi.default_params(0, rate=float(rate))
    calls = [20, 257, 600, 33, 700, 5, 64, 421]
    assert sum(calls) == 2100
    ref_pts = pts
    for k, n in enumerate(calls):
        ctx.iterate(p, n)
        ref = O.iterate(img, ref_pts, tris, 0, ratio, float(rate), n, literal=False)
        _compare(ctx, ref, 0, "call %d (%d grad-iters)" % (k, n))
        ref_pts = ref["points"]
    assert ctx.info(capi.INFO_PERSIST_ITERS) == 2100 and ctx.info(capi.INFO_PERSIST_FAILURES) == 0
    assert ctx.info(capi.INFO_REPLANS) >= 1, "no plan was cut again: the mesh did not drift"
    assert ctx.info(capi.INFO_WARM_LAUNCHES) >= 3
    moved = np.abs((ref_pts - pts) * np.array([W / 2 / ratio, H / 2], np.float32)).max()
    assert moved > 4.0, moved
    ctx.close()


def test_sixteen_grad_iters_from_an_old_state_at_the_metric_size():
    """2048^2 / 3000: 4096 grad-iters on the device (re-cuts, re-plans, chunked launches), then the positions go to the oracle and 16 more
    grad-iters in two calls are compared -- the state the benchmark's long calls run in, not the fresh one"""
    ctx, sweep, pts, tris, ratio, colors = _setup(2048, 2048, 3000, 0, 0.1)
    p = capi.default_params(0)
    ctx.iterate(p, 4096)
    old = ctx.retrieve(capi.BUF_POINTS)
    assert not np.array_equal(old, pts)
    ctx.iterate(p, 9)
    ctx.iterate(p, 7)
    ref = O.iterate(sweep, old, tris, 0, ratio, RATE[0], 16, literal=False)
    _compare(ctx, ref, 0, "16 grad-iters behind 4096")
    assert ctx.info(capi.INFO_PERSIST_ITERS) == 4096 + 16 and ctx.info(capi.INFO_PERSIST_FAILURES) == 0
    ctx.close()
