"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit-exact.

Integer buffers (`colnum`, `colacc`, `tenergy`, `gradient`, moments) must be identical; vertex
positions are float32 produced by the same sequence of IEEE operations (shift.cs:45) and must be
BIT-identical too (tolerance 0 ulp, stated here; see DESIGN.md "Numerics").
"""
import os

import numpy as np
import pytest

from oracle import oracle as O
from tpose_amd import capi, synth
from util import RATE, case

pytestmark = pytest.mark.gpu

CASES = [
    (64, 48, (6, 4)), (97, 61, (6, 4)), (300, 200, (15, 5)), (257, 131, (6, 4)),
    (640, 480, (50, 30)), (128, 32, None), (129, 33, None), (1011, 674, (15, 5)),
]


def run_piecewise(ctx, flavour, slot, rate):
    ctx.accumulate(flavour, slot)
    ctx.energy(flavour)
    out = dict(ten=ctx.retrieve(capi.BUF_TENERGY), cn=ctx.retrieve(capi.BUF_COLNUM),
               ca=ctx.retrieve(capi.BUF_COLACC), mom=ctx.retrieve(capi.BUF_MOMENTS))
    ctx.shift(rate)
    out["gr"] = ctx.retrieve(capi.BUF_GRADIENT)
    out["points"] = ctx.retrieve(capi.BUF_POINTS)
    return out


@pytest.mark.parametrize("W,H,grid", CASES)
@pytest.mark.parametrize("flavour", [0, 1])
def test_single_iteration_matches_oracle(W, H, grid, flavour):
    img, imgB, pts, tris, ratio, colors = case(W, H, grid)
    ctx = capi.Context(0, W, H)
    ctx.set_image(capi.IMAGE_A, img)
    ctx.set_image(capi.IMAGE_B, imgB)
    ctx.upload(pts, tris, colors if flavour == 1 else None)
    slot = capi.IMAGE_B if flavour == 1 else capi.IMAGE_A
    got = run_piecewise(ctx, flavour, slot, RATE[flavour])
    sweep = imgB if flavour == 1 else img
    ref = O.iterate(sweep, pts, tris, flavour, ratio, RATE[flavour], 1,
                    colors=colors if flavour == 1 else None, literal=True)
    mom = O.moments(sweep, pts, tris, O.dp(flavour, tris.shape[0]), ratio)
    assert np.array_equal(got["mom"], mom)
    assert np.array_equal(got["cn"], ref["cn"])
    assert np.array_equal(got["ten"], ref["ten"])
    assert np.array_equal(got["ca"], ref["ca"])
    assert np.array_equal(got["gr"], ref["gr"])
    assert np.array_equal(got["points"].view(np.uint32), ref["points"].view(np.uint32))
    ctx.close()


@pytest.mark.parametrize("flavour", [0, 1])
@pytest.mark.parametrize("iters", [1, 10, 50])
def test_fused_iterations_match_oracle(flavour, iters):
    W, H, grid = 300, 200, (15, 5)
    img, imgB, pts, tris, ratio, colors = case(W, H, grid)
    sweep = imgB if flavour == 1 else img
    ctx = capi.Context(0, W, H)
    ctx.set_image(capi.IMAGE_A, img)
    ctx.set_image(capi.IMAGE_B, imgB)
    ctx.upload(pts, tris, colors if flavour == 1 else None)
    ctx.iterate(capi.default_params(flavour), iters)
    ref = O.iterate(sweep, pts, tris, flavour, ratio, RATE[flavour], iters,
                    colors=colors if flavour == 1 else None, literal=False)
    assert np.array_equal(ctx.retrieve(capi.BUF_TENERGY), ref["ten"])
    assert np.array_equal(ctx.retrieve(capi.BUF_COLNUM), ref["cn"])
    assert np.array_equal(ctx.retrieve(capi.BUF_GRADIENT), ref["gr"])
    assert np.array_equal(ctx.retrieve(capi.BUF_POINTS).view(np.uint32), ref["points"].view(np.uint32))
    ctx.close()


def test_random_soup_and_ties():
    """Arbitrary (overlapping, inverted, degenerate, out-of-domain, lattice-aligned) triangles."""
    W, H = 200, 150
    img = synth.voronoi_raster(W, H, seed=3, sites=10)
    ratio = float(np.float32(W) / np.float32(H))
    rng = np.random.default_rng(0)
    ctx = capi.Context(0, W, H)
    ctx.set_image(capi.IMAGE_A, img)
    for trial in range(12):
        NP = 30
        pts = (rng.random((NP, 2)).astype(np.float32) * 2 - 1) * np.float32(1.3)
        pts[:, 0] *= np.float32(ratio)
        if trial % 3 == 0:
            pts = (np.round(pts * 8) / 8).astype(np.float32)
        tris = np.zeros((40, 4), np.int32)
        tris[:, :3] = rng.integers(0, NP, (40, 3))
        dp = [0.05, 0.0078125, 0.3][trial % 3]
        ctx.upload(pts, tris)
        ctx.set_dp(dp)
        ctx.accumulate(0, capi.IMAGE_A)
        ctx.energy(0)
        mom = O.moments(img, pts, tris, dp, ratio)
        assert np.array_equal(ctx.retrieve(capi.BUF_MOMENTS), mom), "trial %d" % trial
    ctx.close()


def test_full_size_properties():
    """BASELINE.json metric size (2048^2, 3000 triangles): size-independent properties.
    The base variants of a triangulation tile the raster exactly once (watertight top-left rule), so
    sum(cn[0:NT]) == W*H and the base colour sums equal the image's channel sums."""
    W = H = 2048
    img, pts, tris, he, ratio = synth.workload(W, H, 3000)
    ctx = capi.Context(0, W, H)
    ctx.set_image(capi.IMAGE_A, img)
    ctx.upload(pts, tris)
    ctx.accumulate(0, capi.IMAGE_A)
    ctx.energy(0)
    NT = tris.shape[0]
    cn = ctx.retrieve(capi.BUF_COLNUM)
    ca = ctx.retrieve(capi.BUF_COLACC)
    mom = ctx.retrieve(capi.BUF_MOMENTS)
    assert int(cn[:NT].sum()) == W * H
    assert np.array_equal(ca[:NT, :3].astype(np.int64).sum(axis=0), img[:, :, :3].astype(np.int64).sum(axis=(0, 1)))
    q = (img[:, :, :3].astype(np.int64) ** 2).sum()
    assert int(mom[:NT, 5].sum()) == int(q)
    # spot-check 40 variants against the oracle restricted to those triangles
    sel = np.arange(0, NT, NT // 40)[:40]
    sub = tris[sel]
    om = O.moments(img, pts, sub, O.dp(0, NT), ratio).reshape(13, len(sel), 6)
    gm = mom.reshape(13, NT, 6)[:, sel]
    assert np.array_equal(om, gm)
    # iterate: energy must not increase wildly and points stay finite / inside the domain
    ctx.iterate(capi.default_params(0), 32)
    p = ctx.retrieve(capi.BUF_POINTS)
    assert np.isfinite(p).all()
    ctx.close()


def test_device_walker_floor_is_exact():
    """The division-free edge walker on the GPU (v_rcp_f64 + Newton) against exact integer floors."""
    rng = np.random.default_rng(11)
    n = 200000
    d = rng.integers(1, 1 << 24, n)
    d[: n // 8] = rng.choice([1, 2, 3, 255, 256, 257, (1 << 24) - 1], n // 8)
    step = rng.integers(-(1 << 24) + 1, 1 << 24, n)
    N0 = rng.integers(-(1 << 40), 1 << 40, n)
    k = n // 2
    N0[:k] = d[:k] * rng.integers(-(1 << 15), 1 << 15, k) + rng.integers(-1, 2, k)   # knife edges
    step[: k // 2] = d[: k // 2] * rng.integers(-3, 4, k // 2)
    ctx = capi.Context(0, 64, 64)
    got = ctx.selftest_walker(N0, step, d).astype(np.int64)
    exact = (N0[:, None] + np.arange(32)[None, :] * step[:, None].astype(np.int64)) // d[:, None]
    inside = np.abs(N0 // d) <= (1 << 30)
    assert np.array_equal(got[inside], exact[inside])
    far = ~inside
    assert np.all((got[far] > (1 << 28)) == (exact[far] > 0)) and np.all(np.abs(got[far]) > (1 << 28))
    ctx.close()


def test_device_whole_line_walker_is_exact():
    """The whole-line 24.40 walker of round 2 (one set-up per line and iteration; tiles derive their 32.32 walker
    from it) on the GPU against exact integer arithmetic: up to 300 rows of every line, rasters up to 16384 rows,
    the full coordinate range, lattice-aligned and nearly horizontal / vertical lines."""
    rng = np.random.default_rng(21)
    n, rows = 60000, 300
    lo, hi = -(1 << 22), 1 << 23
    ends = rng.integers(lo, hi + 1, (n, 4)).astype(np.int64)
    H = rng.choice([16384, 4096, 2048, 600, 97], n).astype(np.int64)
    k = n // 4
    ends[:k] = rng.integers(-100 * 256, 2148 * 256, (k, 4))                       # raster-sized
    ends[k:2 * k] = 256 * rng.integers(-50, 4200, (k, 4)) + 128                     # pixel centres: ties everywhere
    ends[2 * k:2 * k + k // 2, 2] = ends[2 * k:2 * k + k // 2, 0] + rng.integers(-5, 6, k // 2)   # nearly vertical
    ends[2 * k + k // 2:3 * k, 3] = ends[2 * k + k // 2:3 * k, 1] + rng.integers(1, 700, k // 2)  # nearly horizontal
    ends = np.clip(ends, lo, hi)
    ctx = capi.Context(0, 64, 64)
    got = ctx.selftest_line(ends, H, rows).astype(np.int64)
    ctx.close()
    Xa, Ya, Xb, Yb = ends.T
    swap = Ya > Yb
    Xt, Yt = np.where(swap, Xb, Xa), np.where(swap, Yb, Ya)
    Xq, Yq = np.where(swap, Xa, Xb), np.where(swap, Ya, Yb)
    dy, dx = Yq - Yt, Xq - Xt
    ra = np.maximum((Yt - 128 + 255) >> 8, 0)
    rb = np.minimum(((Yq - 128 + 255) >> 8) - 1, H - 1)
    live = (dy > 0) & (ra <= rb)
    assert np.array_equal(got[~live, 0] > got[~live, 1], np.ones((~live).sum(), bool))
    assert np.array_equal(got[live, 0], ra[live]) and np.array_equal(got[live, 1], rb[live])
    # first column c with (256 c + 128 - Xt) dy >= dx (256 r + 128 - Yt); python ints: no overflow
    idx = np.nonzero(live)[0]
    r = ra[idx, None] + np.arange(rows)[None, :]
    ok = r <= rb[idx, None]
    num = dx[idx, None].astype(object) * (256 * r + 128 - Yt[idx, None]).astype(object) - \
        (128 - Xt[idx, None]).astype(object) * dy[idx, None].astype(object)
    den = (256 * dy[idx, None]).astype(object)
    exact = -((-num) // den)   # ceil
    assert np.array_equal(got[idx, 2:][ok], exact[ok].astype(np.int64))


@pytest.mark.parametrize("persistent", [0, 1])
def test_fast_moving_vertices_match_oracle(persistent):
    """A high step rate makes vertices travel many pixels per grad-iter: the persistent kernel's lanes find their cached
    table records stale almost every time (and lines outgrow the rows a lane keeps); the two-kernel path sees lines far
    longer than at upload.  Either way the oracle's bits."""
    W, H, grid = 300, 200, (15, 5)
    img, imgB, pts, tris, ratio, colors = case(W, H, grid)
    ctx = capi.Context(0, W, H)
    ctx.set_persistent(persistent)
    ctx.set_image(capi.IMAGE_A, img)
    ctx.upload(pts, tris)
    rate = 0.0004
    params = capi.default_params(0, rate=rate)
    ctx.iterate(params, 40)
    ref = O.iterate(img, pts, tris, 0, ratio, rate, 40, literal=False)
    assert np.array_equal(ctx.retrieve(capi.BUF_TENERGY), ref["ten"])
    assert np.array_equal(ctx.retrieve(capi.BUF_POINTS).view(np.uint32), ref["points"].view(np.uint32))
    moved = np.abs(ref["points"] - pts).max() * H / 2
    assert moved > 3.0, moved                      # the test really moves vertices
    ctx.close()


def test_short_calls_replan_on_the_side_and_keep_the_bits():
    """A loop of calls of a few hundred grad-iters on a drifting mesh: the plan of the persistent launches is cut again
    on another host thread after a call (never inside one: such a call has no second chunk), a later call installs it,
    and nothing of that shows in the results -- positions and energies bit-equal to the two-kernel path after every call."""
    W, H, NT = 640, 480, 3000
    img, pts, tris, he, ratio = synth.workload(W, H, NT, contrast=0.3)
    params = capi.default_params(0)
    ctxs = []
    for persistent in (1, 0):
        c = capi.Context(0, W, H)
        c.set_persistent(persistent)
        c.set_image(capi.IMAGE_A, img)
        c.upload(pts, tris)
        ctxs.append(c)
    a, b = ctxs
    import time
    for call in range(14):
        a.iterate(params, 300)
        b.iterate(params, 300)
        pa, pb = a.retrieve(capi.BUF_POINTS), b.retrieve(capi.BUF_POINTS)
        assert np.array_equal(pa.view(np.uint32), pb.view(np.uint32)), call
        assert np.array_equal(a.retrieve(capi.BUF_TENERGY), b.retrieve(capi.BUF_TENERGY)), call
        time.sleep(0.01)   # (the cut takes a few milliseconds: let a later call find it finished)
    assert a.info(capi.INFO_PERSIST_ITERS) == 14 * 300
    moved = np.abs(a.retrieve(capi.BUF_POINTS) - pts).max() * H / 2
    assert moved > 2.0, moved
    assert a.info(capi.INFO_REPLANS) >= 1, "the mesh drifted %.1f px and no plan was cut on the side" % moved
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("W,H", [(1011, 674), (2048, 2048)])
def test_two_triangle_start_state_large_raster(W, H):
    """The reference's 2-triangle start state on big rasters: lines of hundreds to thousands of rows
    (one line per workgroup in k_lines) against the oracle's moments."""
    img = synth.voronoi_raster(W, H, seed=11, sites=24)
    ratio = float(np.float32(W) / np.float32(H))
    pts, tris, _ = synth.two_triangle(ratio)
    ctx = capi.Context(0, W, H)
    ctx.set_image(capi.IMAGE_A, img)
    ctx.upload(pts, tris)
    ctx.accumulate(0, capi.IMAGE_A)
    ctx.energy(0)
    mom = O.moments(img, pts, tris, O.dp(0, 2), ratio)
    assert np.array_equal(ctx.retrieve(capi.BUF_MOMENTS), mom)
    assert int(ctx.retrieve(capi.BUF_COLNUM)[:2].sum()) == W * H
    ctx.close()


@pytest.mark.parametrize("grid", [(1, 1), (2, 2), (4, 3), (8, 6), (16, 12), (40, 30)])
def test_every_chunking_of_k_lines_matches_oracle(grid):
    """meshes from two triangles to 2400 on a 1024x768 raster: the chunk count per line (and with it the workgroup
    shape of k_lines: three edges per workgroup up to 16 chunks, one line per workgroup beyond) follows the mesh;
    three fused grad-iters, every integer buffer and the positions bit-equal to the oracle"""
    W, H = 1024, 768
    img = synth.voronoi_raster(W, H, seed=21, sites=40)
    ratio = float(np.float32(W) / np.float32(H))
    pts, tris, _ = synth.grid_triangulation(grid[0], grid[1], ratio=ratio)
    ctx = capi.Context(0, W, H)
    ctx.set_image(capi.IMAGE_A, img)
    ctx.upload(pts, tris)
    chunks = ctx.info(1)
    assert chunks >= 1 and (chunks & (chunks - 1)) == 0
    ctx.iterate(capi.default_params(0), 3)
    ref = O.iterate(img, pts, tris, 0, ratio, RATE[0], 3, literal=False)
    assert np.array_equal(ctx.retrieve(capi.BUF_TENERGY), ref["ten"]), chunks
    assert np.array_equal(ctx.retrieve(capi.BUF_COLNUM), ref["cn"])
    assert np.array_equal(ctx.retrieve(capi.BUF_COLACC), ref["ca"])
    assert np.array_equal(ctx.retrieve(capi.BUF_GRADIENT), ref["gr"])
    assert np.array_equal(ctx.retrieve(capi.BUF_POINTS).view(np.uint32), ref["points"].view(np.uint32))
    ctx.close()


def test_widest_raster_saturated_pixels():
    """the packing limits of the prefix table on the device: 16384 columns of (255, 255, 255) -- 22-bit channel
    sums, a 32-bit square sum and a 15-bit parity count filled to the last bit -- and of odd pixels"""
    W, H = 16384, 6
    ratio = float(np.float32(W) / np.float32(H))
    pts, tris, _ = synth.two_triangle(ratio)
    for val in ((255, 255, 255), (255, 254, 0)):
        img = np.zeros((H, W, 4), np.uint8)
        img[..., :3] = val
        img[..., 3] = 255
        ctx = capi.Context(0, W, H)
        ctx.set_image(capi.IMAGE_A, img)
        ctx.upload(pts, tris)
        ctx.accumulate(0, capi.IMAGE_A)
        ctx.energy(0)
        mom = O.moments(img, pts, tris, O.dp(0, 2), ratio)
        assert np.array_equal(ctx.retrieve(capi.BUF_MOMENTS), mom)
        assert int(ctx.retrieve(capi.BUF_COLNUM)[:2].sum()) == W * H
        ctx.close()


def test_batch_config_properties():
    """BASELINE.json batch configuration (4096^2, 12000 triangles): size-independent properties."""
    W = H = 4096
    img, pts, tris, he, ratio = synth.workload(W, H, 12000)
    ctx = capi.Context(0, W, H)
    ctx.set_image(capi.IMAGE_A, img)
    ctx.upload(pts, tris)
    ctx.accumulate(0, capi.IMAGE_A)
    ctx.energy(0)
    NT = tris.shape[0]
    mom = ctx.retrieve(capi.BUF_MOMENTS)
    assert int(mom[:NT, 0].sum()) == W * H
    assert np.array_equal(mom[:NT, 2:5].sum(axis=0), img[:, :, :3].astype(np.int64).sum(axis=(0, 1)))
    sel = np.arange(0, NT, NT // 24)[:24]
    om = O.moments(img, pts, tris[sel], O.dp(0, NT), ratio).reshape(13, len(sel), 6)
    assert np.array_equal(om, mom.reshape(13, NT, 6)[:, sel])
    ctx.iterate(capi.default_params(0), 20)
    assert np.isfinite(ctx.retrieve(capi.BUF_POINTS)).all()
    ctx.close()


def _check_moments(W, H, pts, tris, dp=None, seed=5):
    img = synth.voronoi_raster(W, H, seed=seed, sites=7)
    ratio = float(np.float32(W) / np.float32(H))
    ctx = capi.Context(0, W, H)
    ctx.set_image(capi.IMAGE_A, img)
    ctx.upload(pts, tris)
    if dp is not None:
        ctx.set_dp(dp)
    ctx.accumulate(0, capi.IMAGE_A)
    ctx.energy(0)
    d = O.dp(0, tris.shape[0]) if dp is None else dp
    got = ctx.retrieve(capi.BUF_MOMENTS)
    ctx.close()
    assert np.array_equal(got, O.moments(img, pts, tris, d, ratio))
    return got


def test_edge_cases_small_and_degenerate():
    """ragged / degenerate inputs: 1x1 and 1-pixel-wide rasters, exact tile multiples, a single
    triangle, triangles entirely outside the domain, zero-area and repeated triangles"""
    for W, H in [(1, 1), (1, 40), (300, 1), (128, 32), (256, 64), (129, 33)]:
        ratio = float(np.float32(W) / np.float32(H))
        pts, tris, _ = synth.two_triangle(ratio)
        _check_moments(W, H, pts, tris)
    W, H = 150, 90
    ratio = float(np.float32(W) / np.float32(H))
    one = (np.array([[-1.0, -0.5], [0.2, 0.9], [1.1, -0.7]], np.float32) * np.array([ratio / 1.7, 1], np.float32)).astype(np.float32)
    _check_moments(W, H, one, np.array([[0, 1, 2, 0]], np.int32))
    outside = (one + np.array([5.0 * ratio, 0.0], np.float32)).astype(np.float32)       # far right of the domain
    m = _check_moments(W, H, outside, np.array([[0, 1, 2, 0]], np.int32))
    assert int(m[:, 0].sum()) == 0
    above = (one + np.array([0.0, 4.0], np.float32)).astype(np.float32)
    assert int(_check_moments(W, H, above, np.array([[0, 1, 2, 0]], np.int32))[:, 0].sum()) == 0
    pts = np.array([[-1, -1], [-1, 1], [1, -1], [1, 1], [0, 0], [0, 0], [0.5, 0.5]], np.float32)
    pts[:, 0] *= np.float32(ratio)
    tris = np.array([[0, 1, 2, 0], [0, 1, 2, 0], [4, 5, 6, 0], [4, 4, 4, 0], [2, 1, 3, 0], [3, 1, 2, 0]], np.int32)
    _check_moments(W, H, pts, tris, dp=0.11)


def test_wide_raster_many_tile_columns():
    """4100 x 40: 33 tile columns (one partial), 2 tile rows -- exercises the static table widely"""
    W, H = 4100, 40
    ratio = float(np.float32(W) / np.float32(H))
    pts, tris, _ = synth.grid_triangulation(15, 5, ratio=ratio)
    _check_moments(W, H, pts, tris)
    _check_moments(W, H, pts, tris, dp=0.6)


@pytest.mark.parametrize("kind", ["nan", "inf", "huge", "allsame", "concentrated"])
def test_hostile_vertex_sets_match_oracle(kind):
    """Far-away, coincident or crowded vertices, NaN / inf at the metric size must neither fault nor fail, and the results
    still match the oracle bit for bit."""
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
    else:  # the whole mesh squeezed into ~200 x 100 pixels: thousands of edges per tile
        bad[4:] = bad[4:] * np.float32(0.1) * np.array([1.0, 0.5], np.float32) + np.float32(0.3)
    iters = 3
    ctx = capi.Context(0, W, H)
    ctx.set_image(capi.IMAGE_A, img)
    ctx.upload(bad, tris, None)
    p = capi.default_params(capi.TRIANGULATE)
    ctx.iterate(p, iters)
    ctx.synchronize()
    got_ten, got_cn = ctx.retrieve(capi.BUF_TENERGY), ctx.retrieve(capi.BUF_COLNUM)
    got_pts = ctx.retrieve(capi.BUF_POINTS)
    if kind in ("huge", "allsame", "concentrated"):
        ref = O.iterate(img, bad, tris, O.TRIANGULATE, ratio, RATE[0], iters, literal=False)
        assert np.array_equal(got_ten, ref["ten"]) and np.array_equal(got_cn, ref["cn"])
        assert np.array_equal(got_pts.view(np.uint32), ref["points"].view(np.uint32))
    # the same context, sane input again: results match a fresh context bit for bit
    ctx.upload(pts, tris, None)
    ctx.iterate(p, 2)
    got = ctx.retrieve(capi.BUF_POINTS)
    fresh = capi.Context(0, W, H)
    fresh.set_image(capi.IMAGE_A, img)
    fresh.upload(pts, tris, None)
    fresh.iterate(p, 2)
    want = fresh.retrieve(capi.BUF_POINTS)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    ctx.close(); fresh.close()


def _brute_picture(pts, tris, W, H, ratio, colours):
    """every pixel takes the colour of the base triangle whose coverage mask holds it (tests/brute.py)"""
    import brute
    pic = np.zeros((H, W, 4), np.uint8)
    pic[:, :, 3] = 255
    owners = np.zeros((H, W), np.int32)
    for t in range(tris.shape[0]):
        m = brute.coverage_mask(brute.variant_xy(pts, tris[t], 0, 0.0, ratio, W, H), W, H)
        owners += m
        if colours[t] is not None:
            pic[m, :3] = colours[t]
    return pic, owners


def test_render_flat_shaded_picture():
    """tp_render (display pass, triangle.fs mode 2 / software/view): pixel ownership is the sweep's
    coverage rule; colours are the averages of the last sweep or the uploaded ones"""
    W, H = 300, 200
    img, imgB, pts, tris, ratio, stored = case(W, H, (15, 5))
    ctx = capi.Context(0, W, H)
    ctx.set_image(capi.IMAGE_A, img)
    ctx.upload(pts, tris, stored)
    got_stored = ctx.render(capi.RENDER_STORED)
    want, owners = _brute_picture(pts, tris, W, H, ratio, [c[:3] for c in stored])
    assert owners.min() == 1 and owners.max() == 1  # a triangulation of the domain tiles the raster
    assert np.array_equal(got_stored, want)
    # averages of the sweep: reference display divides in float and the framebuffer rounds to nearest
    ctx.upload(pts, tris, None)
    ctx.accumulate(capi.TRIANGULATE, capi.IMAGE_A)
    ctx.energy(capi.TRIANGULATE)
    ca, cn = ctx.retrieve(capi.BUF_COLACC)[: tris.shape[0]], ctx.retrieve(capi.BUF_COLNUM)[: tris.shape[0]]
    cols = []
    for t in range(tris.shape[0]):
        if cn[t] == 0:
            cols.append(None)
            continue
        f = (ca[t, :3].astype(np.float32) / np.float32(cn[t])) / np.float32(255)
        cols.append(np.floor(np.clip(f, 0, 1) * np.float32(255) + np.float32(0.5)).astype(np.uint8))
    want, _ = _brute_picture(pts, tris, W, H, ratio, cols)
    assert np.array_equal(ctx.render(capi.RENDER_AVERAGE), want)
    # a morphed mesh (software/view draws mix(points, originpoints, s)): positions for this picture only
    moved = pts.copy()
    moved[4:] += np.float32(0.01)
    want, _ = _brute_picture(moved, tris, W, H, ratio, cols)
    assert np.array_equal(ctx.render(capi.RENDER_AVERAGE, points=moved), want)
    assert np.array_equal(ctx.retrieve(capi.BUF_POINTS), pts)
    ctx.close()


def test_batched_readback_equals_single_readbacks():
    """tp_retrieve_many: the reference's four per-frame readbacks with one wait"""
    W, H = 300, 200
    img, imgB, pts, tris, ratio, colors = case(W, H, (15, 5))
    ctx = capi.Context(0, W, H)
    ctx.set_image(capi.IMAGE_A, img)
    ctx.upload(pts, tris, None)
    ctx.iterate(capi.default_params(capi.TRIANGULATE), 3)
    whats = [capi.BUF_TENERGY, capi.BUF_PENERGY, capi.BUF_COLNUM, capi.BUF_POINTS, capi.BUF_COLACC, capi.BUF_GRADIENT]
    many = ctx.retrieve_many(whats)
    for w, a in zip(whats, many):
        b = ctx.retrieve(w)
        assert a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))
    assert not many[1].any()  # penergy is dead in the reference: always zero
    ctx.close()


def test_fused_path_clamps_vertices_no_triangle_uses():
    """shift.cs runs for every vertex i >= 4: a vertex outside the domain that no triangle references is still
    clamped to the domain edge (found by tools/soak.py)"""
    W, H = 233, 101
    img = synth.voronoi_raster(W, H, seed=11, sites=9)
    ratio = float(np.float32(W) / np.float32(H))
    pts = np.array([[-ratio, -1], [-ratio, 1], [ratio, -1], [ratio, 1], [0.1, 0.2], [0.5, -0.3], [-0.4, 0.1],
                    [3.0, 0.5], [-0.2, -7.0], [0.3, 0.3]], np.float32)   # 7, 8 outside and unused; 9 inside and unused
    tris = np.array([[4, 5, 6, 0], [0, 4, 6, 0]], np.int32)
    ctx = capi.Context(0, W, H)
    ctx.set_image(capi.IMAGE_A, img)
    for flavour in (0, 1):
        colors = np.array([[10, 20, 30, 1], [200, 100, 50, 1]], np.int32) if flavour else None
        ctx.upload(pts, tris, colors)
        p = capi.default_params(flavour, image_slot=capi.IMAGE_A)
        ctx.iterate(p, 3)
        ref = O.iterate(img, pts, tris, flavour, ratio, RATE[flavour], 3, colors=colors, literal=False)
        got = ctx.retrieve(capi.BUF_POINTS)
        assert np.array_equal(got.view(np.uint32), ref["points"].view(np.uint32))
        assert got[7, 0] == np.float32(ratio) and got[8, 1] == np.float32(-1.0) and np.array_equal(got[9], pts[9])
    ctx.close()


def test_randomised_soak_sample():
    """a slice of tools/soak.py (random sizes, soups, grids, flavours, dp, margins) as a regular test"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak.py"), "36", "99"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_cached_graphs_follow_ratio_and_dp():
    """captured graphs bake RATIO and dp: changing tpose::RATIO, or the dp of the params, between two tp_iterate
    calls must not replay a stale graph"""
    W, H = 300, 200
    img, imgB, pts, tris, ratio, colors = case(W, H, (15, 5))
    ctx = capi.Context(0, W, H)
    ctx.set_image(capi.IMAGE_A, img)
    for r, dp in ((ratio, 0.0), (1.2, 0.0), (1.2, 0.013), (ratio, 0.013)):
        ctx.set_ratio(r)
        ctx.upload(pts, tris, None)
        p = capi.default_params(capi.TRIANGULATE, dp=dp)
        ctx.iterate(p, 32)  # two graph replays
        ref = O.iterate(img, pts, tris, 0, r, RATE[0], 32, dp_=dp if dp > 0 else None, literal=False)
        assert np.array_equal(ctx.retrieve(capi.BUF_POINTS).view(np.uint32), ref["points"].view(np.uint32)), (r, dp)
    ctx.close()


def test_device_resident_raster_with_padded_stride():
    """tp_set_image_device: the raster already lives in HBM, rows padded (allocated through the HIP runtime the
    library itself uses -- PyTorch-ROCm bundles its own runtime and must be imported first when both are used)"""
    import ctypes
    W, H = 301, 97
    img, imgB, pts, tris, ratio, colors = case(W, H, (9, 4))
    ctx = capi.Context(0, W, H)
    hip = ctypes.CDLL("libamdhip64.so.7" if "torch" not in __import__("sys").modules else "libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy2D.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    stride = (W + 13) * 4
    dev = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(dev), stride * H) == 0
    assert hip.hipMemcpy2D(dev, stride, img.ctypes.data, W * 4, W * 4, H, 1) == 0  # hipMemcpyHostToDevice
    ctx.set_image_device(capi.IMAGE_A, dev.value, stride)
    ctx.upload(pts, tris, None)
    ctx.iterate(capi.default_params(capi.TRIANGULATE), 5)
    ref = O.iterate(img, pts, tris, 0, ratio, RATE[0], 5, literal=False)
    assert np.array_equal(ctx.retrieve(capi.BUF_TENERGY), ref["ten"])
    assert np.array_equal(ctx.retrieve(capi.BUF_POINTS).view(np.uint32), ref["points"].view(np.uint32))
    ctx.close()
    hip.hipFree(dev)


def test_two_host_threads_two_contexts():
    """contexts are independent: two host threads drive one each on the same GPU at the same time"""
    import threading
    W, H = 640, 360
    results = {}

    def work(k):
        img, imgB, pts, tris, ratio, colors = case(W, H, (20 + 3 * k, 11), seed=50 + k)
        ctx = capi.Context(0, W, H)
        ctx.set_image(capi.IMAGE_A, img)
        ctx.upload(pts, tris, None)
        p = capi.default_params(capi.TRIANGULATE)
        for rep in range(20):
            ctx.iterate(p, 17)
        got = (ctx.retrieve(capi.BUF_TENERGY), ctx.retrieve(capi.BUF_POINTS))
        ctx.close()
        results[k] = (got, (img, pts, tris, ratio))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for k in range(2):
        (ten, points), (img, pts, tris, ratio) = results[k]
        ref = O.iterate(img, pts, tris, 0, ratio, RATE[0], 340, literal=False)
        assert np.array_equal(ten, ref["ten"]) and np.array_equal(points.view(np.uint32), ref["points"].view(np.uint32))


def _reference_loop(ctx, params, NT, threshold, max_frames, toterr=1.0):
    """the reference's loop: one frame, read tenergy back, tpose::geterr in float32 ascending t (triangulation.hpp:653-674)"""
    tot = np.float32(toterr)
    frames = 0
    rel = np.float32(0)
    while frames < max_frames:
        ctx.iterate(params, 1)
        frames += 1
        ten = ctx.retrieve(capi.BUF_TENERGY)[:NT]
        newerr = np.float32(0)
        for v in ten.astype(np.float32):
            newerr = np.float32(newerr + v)
        with np.errstate(all="ignore"):
            rel = np.float32(np.float32(tot - newerr) / tot)
        tot = newerr
        if float(abs(rel)) < threshold:
            break
    return frames, float(tot), float(rel)


@pytest.mark.parametrize("flavour,threshold,cap", [(0, 1e-4, 400), (1, 1e-6, 300), (0, 1e-9, 70), (0, 1e-4, 3)])
@pytest.mark.parametrize("persistent", [1, 0])
def test_iterate_until_is_the_reference_loop(flavour, threshold, cap, persistent):
    """tp_iterate_until against the reference's frame loop with a read-back and geterr after every frame: the same number
    of frames, the same running total, and the same buffers and positions afterwards -- bit for bit"""
    W, H, grid = 300, 200, (15, 5)
    img = synth.photo_contrast(synth.voronoi_raster(W, H, seed=7, sites=12), 0.3)
    imgB = synth.displaced_raster(img, amp=6.0)
    ratio = float(np.float32(W) / np.float32(H))
    pts, tris, _ = synth.grid_triangulation(grid[0], grid[1], ratio=ratio)
    colors = synth.mean_colors(img, pts, tris, ratio)
    NT = tris.shape[0]
    res = []
    for mode in ("loop", "until"):
        ctx = capi.Context(0, W, H)
        ctx.set_persistent(persistent)
        ctx.set_image(capi.IMAGE_A, img)
        ctx.set_image(capi.IMAGE_B, imgB)
        ctx.upload(pts, tris, colors if flavour else None)
        p = capi.default_params(flavour)
        out = []
        tot = 1.0
        for leg in range(2):   # two legs: the running total carries over
            if mode == "loop":
                ctx.set_persistent(0)
                n, tot, rel = _reference_loop(ctx, p, NT, threshold, cap, tot)
            else:
                n, tot, rel = ctx.iterate_until(p, cap, threshold, tot)
            out.append((n, np.float32(tot).view(np.uint32), np.float32(rel).view(np.uint32)))
        res.append((out, ctx.retrieve(capi.BUF_POINTS), ctx.retrieve(capi.BUF_TENERGY), ctx.retrieve(capi.BUF_COLNUM),
                    ctx.retrieve(capi.BUF_GRADIENT)))
        ctx.close()
    assert res[0][0] == res[1][0], (res[0][0], res[1][0])
    assert np.array_equal(res[0][1].view(np.uint32), res[1][1].view(np.uint32))
    for k in (2, 3, 4):
        assert np.array_equal(res[0][k], res[1][k])


def test_two_contexts_iterate_concurrently_on_one_gpu():
    """two contexts driven from two threads: their persistent launches compete for the same CUs (a launch whose
    workgroups are not all resident gives up and its grad-iters are run again on the two-kernel path) -- results exact"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "two_contexts.py")], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("flavour", [0, 1])
def test_frame_mirror_returns_what_the_device_holds(flavour):
    """single frames (tp_iterate(1), or energy + shift) leave the first NT + 64 entries of `tenergy` / `colnum` and the points
    in pinned memory as well, and small read-backs right behind them are served from there: every such read-back equals the
    copy from device memory (a batch with a buffer the mirror does not hold takes the copy path), also across the calls that
    must invalidate it"""
    W, H = 300, 200
    img, imgB, pts, tris, ratio, colors = case(W, H, (15, 5))
    ctx = capi.Context(0, W, H)
    ctx.set_image(capi.IMAGE_A, img)
    ctx.set_image(capi.IMAGE_B, imgB)
    ctx.upload(pts, tris, colors if flavour else None)
    p = capi.default_params(flavour)
    NT = tris.shape[0]

    def check(tag):
        small = [ctx.retrieve(capi.BUF_TENERGY, NT + 2)[: NT + 2], ctx.retrieve(capi.BUF_COLNUM, NT + 2)[: NT + 2], ctx.retrieve(capi.BUF_POINTS)]
        full = ctx.retrieve_many([capi.BUF_TENERGY, capi.BUF_COLNUM, capi.BUF_POINTS, capi.BUF_COLACC])   # (colacc: the copy path)
        assert np.array_equal(small[0], full[0][: NT + 2]), tag
        assert np.array_equal(small[1], full[1][: NT + 2]), tag
        assert np.array_equal(small[2].view(np.uint32), full[2].view(np.uint32)), tag

    slot = capi.IMAGE_B if flavour else capi.IMAGE_A
    for rep in range(3):
        ctx.iterate(p, 1); check("frame")
        ctx.iterate(p, 1); ctx.iterate(p, 1); check("two frames")
        ctx.accumulate(flavour, slot); ctx.energy(flavour); check("energy")
        ctx.shift(RATE[flavour]); check("energy + shift")
        ctx.iterate(p, 1); ctx.accumulate(flavour, slot); check("frame, then a sweep")
        ctx.iterate(p, 1); ctx.iterate(p, 7); check("frame, then a persistent call")
        ctx.iterate(p, 1)
        moved = ctx.retrieve(capi.BUF_POINTS)
        ctx.upload(moved, tris, colors if flavour else None); check("frame, then an upload")
        ctx.iterate(p, 1); ctx.set_dp(0.02); ctx.accumulate(flavour, slot); ctx.energy(flavour); check("another dp")
        ctx.set_dp(0.0)
    # and the frames are the oracle's
    ctx.upload(pts, tris, colors if flavour else None)
    for k in range(5):
        ctx.iterate(p, 1)
        got = ctx.retrieve(capi.BUF_POINTS)
    ref = O.iterate(imgB if flavour else img, pts, tris, flavour, ratio, RATE[flavour], 5, colors=colors if flavour else None, literal=False)
    assert np.array_equal(got.view(np.uint32), ref["points"].view(np.uint32))
    assert np.array_equal(ctx.retrieve(capi.BUF_TENERGY, NT)[:NT], ref["ten"][:NT])
    ctx.close()


@pytest.mark.parametrize("size,grid", [((300, 200), (15, 5)), ((674, 449), (10, 8))])
def test_evaluate_triangles_is_the_base_energy(size, grid):
    """tp_evaluate_triangles: the energy and pixel count a triangle WOULD have at the current positions equal what the sweep leaves for
    it as a base variant -- for the triangles of the mesh (any vertex order), for triangles that are not in it (the other diagonal of a
    quad: what the schedule's flip set asks for; checked against the oracle on a mesh that holds them), and for one without area"""
    W, H = size
    img, imgB, pts, tris, ratio, colors = case(W, H, grid)
    ctx = capi.Context(0, W, H)
    ctx.set_image(capi.IMAGE_A, img)
    ctx.upload(pts, tris, None)
    p = capi.default_params(capi.TRIANGULATE)
    ctx.iterate(p, 7)    # (somewhere off the grid)
    moved = ctx.retrieve(capi.BUF_POINTS)
    ref = O.iterate(img, moved, tris, O.TRIANGULATE, ratio, RATE[0], 1, literal=False)   # `ten`, `cn`: the sweep at `moved`
    NT = tris.shape[0]
    e, n = ctx.evaluate_triangles(tris[:, :3])
    assert np.array_equal(e, ref["ten"][:NT]) and np.array_equal(n, ref["cn"][:NT])
    # every displaced variant: entry variant * NT + t of the sweep's buffers (the dp of the context's sweeps: the law at the uploaded NT)
    for i in range(1, 13):
        ei, ni = ctx.evaluate_triangles(tris[:, :3], variants=np.full(NT, i, np.int32))
        assert np.array_equal(ei, ref["ten"][i * NT: (i + 1) * NT]) and np.array_equal(ni, ref["cn"][i * NT: (i + 1) * NT]), i
    # ... and with tp_set_dp in force: the displaced variants follow it
    ctx.set_dp(0.02)
    ref_dp = O.iterate(img, moved, tris, O.TRIANGULATE, ratio, RATE[0], 1, dp_=0.02, literal=False)
    for i in (1, 6, 12):
        ei, ni = ctx.evaluate_triangles(tris[:, :3], variants=np.full(NT, i, np.int32))
        assert np.array_equal(ei, ref_dp["ten"][i * NT: (i + 1) * NT]) and np.array_equal(ni, ref_dp["cn"][i * NT: (i + 1) * NT]), ("dp", i)
    ctx.set_dp(0.0)
    e2, n2 = ctx.evaluate_triangles(tris[:, [2, 0, 1]])          # rotated
    e3, n3 = ctx.evaluate_triangles(tris[:, [1, 0, 2]])          # mirrored
    assert np.array_equal(e2, e) and np.array_equal(e3, e) and np.array_equal(n2, n) and np.array_equal(n3, n)
    # the other diagonal of the quads the grid is made of: triangles (2k, 2k + 1) share an edge
    hyp = []
    for k in range(0, NT - 1, 2):
        a, b = set(tris[k, :3].tolist()), set(tris[k + 1, :3].tolist())
        shared = sorted(a & b)
        if len(shared) != 2:
            continue
        pa, pb = (a - b).pop(), (b - a).pop()
        hyp.append([pa, pb, shared[0]]); hyp.append([pb, pa, shared[1]])
    hyp = np.array(hyp, np.int32)
    assert hyp.shape[0] >= NT // 2
    eh, nh = ctx.evaluate_triangles(hyp)
    tris_h = np.zeros((hyp.shape[0], 4), np.int32); tris_h[:, :3] = hyp
    ref_h = O.iterate(img, moved, tris_h, O.TRIANGULATE, ratio, RATE[0], 1, literal=False)
    assert np.array_equal(eh, ref_h["ten"][: hyp.shape[0]]) and np.array_equal(nh, ref_h["cn"][: hyp.shape[0]])
    ez, nz = ctx.evaluate_triangles(np.array([[5, 5, 9], [7, 7, 7]], np.int32))   # no area: no pixel, no energy
    assert np.array_equal(ez, [0, 0]) and np.array_equal(nz, [0, 0])
    # nothing of the context changed, and bad indices are refused
    assert np.array_equal(ctx.retrieve(capi.BUF_POINTS).view(np.uint32), moved.view(np.uint32))
    with pytest.raises(Exception):
        ctx.evaluate_triangles(np.array([[0, 1, pts.shape[0]]], np.int32))
    ctx.close()


@pytest.mark.gpu
def test_packed_variant_arithmetic_on_the_device():
    """round 5: P6 of k_persist sums line sums as packed 64-bit words and takes the channel averages through one float reciprocal with a
    remainder fix (tp_persist.h: pk_signed_packed, pk_energy_var).  On the device, against the general 64-bit form and against Python integers:
    pixel sets of two colours up to 2^24 pixels, averages that divide exactly (255 n / n), one below and one above, both flavours."""
    rng = np.random.default_rng(5)
    n_cases = 200000
    n = rng.integers(0, (1 << 24) + 1, n_cases).astype(np.int64)
    n[:20000] = rng.integers(1, 5000, 20000)
    n[20000:20010] = [0, 1, 2, (1 << 23) - 1, 1 << 23, (1 << 23) + 1, 1 << 24, (1 << 24) - 1, 3, 255]
    c1 = rng.integers(0, 256, (n_cases, 3)).astype(np.int64); c2 = rng.integers(0, 256, (n_cases, 3)).astype(np.int64)
    c1[::7] = 255; c2[::11] = 0; c2[::13] = 255
    n1 = (rng.random(n_cases) * (n + 1)).astype(np.int64)
    n1[::5] = n[::5]                 # one colour: the average divides exactly
    n1[1::5] = np.maximum(n[1::5] - 1, 0)   # ... one pixel short of it
    n2 = n - n1
    odd1, odd2 = c1.sum(1) & 1, c2.sum(1) & 1
    M = np.stack([n, n1 * odd1 + n2 * odd2, n1 * c1[:, 0] + n2 * c2[:, 0], n1 * c1[:, 1] + n2 * c2[:, 1], n1 * c1[:, 2] + n2 * c2[:, 2],
                  n1 * (c1 ** 2).sum(1) + n2 * (c2 ** 2).sum(1)], 1)
    # three lines with Wa + Wb - Wc = M (fields inside a line sum's: n, n_odd <= 2^24; r, g < 2^32)
    cap = np.array([1 << 24, 1 << 24, (1 << 32) - 1, (1 << 32) - 1, 1 << 32, 1 << 42], np.int64)
    room = cap[None, :] - M
    X = (rng.random(M.shape) * (room // 2 + 1)).astype(np.int64); Y = (rng.random(M.shape) * (room // 2 + 1)).astype(np.int64)
    Ws = [M + X, Y, X + Y]
    sign = np.array([1, 1, -1])[None, :] * np.where(rng.random(n_cases) < 0.5, 1, -1)[:, None]
    perm = np.argsort(rng.random((n_cases, 3)), 1)
    flips_bits = rng.integers(0, 2, (n_cases, 3))
    sums = np.zeros((n_cases, 3, 4), np.uint64)
    meta = np.zeros((n_cases, 8), np.int32)
    for k in range(3):
        Wk = np.choose(perm[:, k][:, None], [Ws[0], Ws[1], Ws[2]])
        sk = np.take_along_axis(sign, perm[:, k][:, None], 1)[:, 0]
        sums[:, k, 0] = (Wk[:, 0] | (Wk[:, 1] << 32)).astype(np.uint64); sums[:, k, 1] = (Wk[:, 2] | (Wk[:, 3] << 32)).astype(np.uint64)
        sums[:, k, 2] = Wk[:, 4].astype(np.uint64); sums[:, k, 3] = Wk[:, 5].astype(np.uint64)
        meta[:, k] = np.where(flips_bits[:, k] == 1, -sk, sk)
        meta[:, 3] |= (flips_bits[:, k] << k).astype(np.int32)
    meta[:, 4] = rng.integers(0, 2, n_cases)
    meta[:, 5:8] = rng.integers(0, 256, (n_cases, 3))
    wild = (rng.random(n_cases) < 0.1) & (meta[:, 4] == 1)
    meta[wild, 6] = rng.integers(-2**31, 2**31, int(wild.sum()))   # a caller's colour outside a byte: the general form
    ctx = capi.Context(0, 64, 64)
    out = ctx.selftest_variant(sums, meta)
    ctx.close()
    assert np.all(out[:, 9] == 1)
    got = out[:, :5].astype(np.int64) & 0xffffffff
    assert np.array_equal(got, M[:, :5])
    assert np.array_equal((out[:, 5].astype(np.int64) & 0xffffffff) | ((out[:, 6].astype(np.int64) & 0xffffffff) << 32), M[:, 5])
    assert np.array_equal(out[:, 7], out[:, 8])
    # ... and what the reference computes (triangle.fs:37-43), in Python integers, for the cases no int32 of it wraps in
    tri = np.nonzero((meta[:, 4] == 0) & (n > 0) & (n < (1 << 23)))[0][:20000]
    for i in tri[:: max(1, len(tri) // 3000)]:
        nn, no, sr, sg, sb, q = (int(x) for x in M[i])
        a = (sr // nn, sg // nn, sb // nn)
        a2 = a[0] ** 2 + a[1] ** 2 + a[2] ** 2
        S = q - 2 * (a[0] * sr + a[1] * sg + a[2] * sb) + nn * a2
        nodd = nn - no if a2 & 1 else no
        want = ((S - nodd) // 2) & 0xffffffff
        assert (int(out[i, 7]) & 0xffffffff) == want, (i, M[i])
