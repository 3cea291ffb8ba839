"""tp_iterate_frames (round 6): the reference's frame loop with the host in it but off the device's critical path -- frames run in chunks
inside persistent launches, the caller sees every frame's base energies and positions in order and decides where the run ends
(software/triangulate/main.cpp:196-346: geterr, then the prune / wide-angle / collapse sweeps, every frame).  Against the frame-by-frame
loop the reference writes -- one grad-iter, four read-backs -- on the HIP path, bit for bit; tests/test_configs.py and tests/test_harness.py
compare the schedules that use it with `-literal` byte for byte."""
import numpy as np
import pytest

from oracle import oracle as O
from tpose_amd import capi, synth
from util import RATE

pytestmark = pytest.mark.gpu


def _mesh(W, H, loose=False):
    img = synth.workload(W, H, 150, contrast=0.3)[0]
    ratio = float(np.float32(W) / np.float32(H))
    pts, tris, _ = synth.grid_triangulation(15, 5, ratio=ratio)
    if loose:   # two vertices no triangle uses, one of them outside the domain: shift.cs:25-43 clamps every i >= 4
        pts = np.vstack([pts, np.array([[0.1, 0.2], [ratio + 0.5, -1.7]], np.float32)])
    return img, pts, tris, ratio


def _ctx(img, pts, tris):
    c = capi.Context(0, img.shape[1], img.shape[0])
    c.set_image(capi.IMAGE_A, img)
    c.upload(pts, tris)
    return c


@pytest.mark.parametrize("loose", [False, True])
def test_every_frame_is_the_frame_by_frame_loop(loose):
    img, pts, tris, ratio = _mesh(300, 200, loose)
    NT = tris.shape[0]
    a, b = _ctx(img, pts, tris), _ctx(img, pts, tris)
    p = capi.default_params(0)
    seen = []
    n = a.iterate_frames(p, 300, lambda k, ten, q: (seen.append((k, ten.copy(), q.copy())), capi.FRAME_GO_ON)[1])
    assert n == 300 and [s[0] for s in seen] == list(range(300))
    assert a.info(capi.INFO_PERSIST_ITERS) >= 256 and a.info(capi.INFO_PERSIST_FAILURES) == 0   # (the frames ran inside persistent launches)
    for k in range(300):
        b.iterate(p, 1)
        ten, q = b.retrieve_many([capi.BUF_TENERGY, capi.BUF_POINTS])
        assert np.array_equal(seen[k][1], ten[:NT]), "tenergy of frame %d" % k
        assert np.array_equal(seen[k][2].view(np.uint32), q.view(np.uint32)), "positions after frame %d" % k
    # the device holds the last frame's positions, and the next call goes on from there
    assert np.array_equal(a.retrieve(capi.BUF_POINTS).view(np.uint32), b.retrieve(capi.BUF_POINTS).view(np.uint32))
    a.iterate(p, 5); b.iterate(p, 5)
    for what in (capi.BUF_TENERGY, capi.BUF_COLNUM, capi.BUF_GRADIENT):
        assert np.array_equal(a.retrieve(what), b.retrieve(what))
    assert np.array_equal(a.retrieve(capi.BUF_POINTS).view(np.uint32), b.retrieve(capi.BUF_POINTS).view(np.uint32))
    a.close(); b.close()


@pytest.mark.parametrize("stop_at,verdict", [(23, capi.FRAME_STOP_REPLAY), (130, capi.FRAME_STOP_REPLAY), (10, capi.FRAME_STOP), (255, capi.FRAME_STOP), (256, capi.FRAME_STOP_REPLAY)])
def test_a_run_ends_where_the_caller_says(stop_at, verdict):
    """the frame that stops the run: replayed (all four buffers as the reference's frame leaves them -- the oracle's) or its positions restored;
    frames the device ran beyond it are dropped"""
    img, pts, tris, ratio = _mesh(300, 200, True)
    a = _ctx(img, pts, tris)
    p = capi.default_params(0)
    n = a.iterate_frames(p, 1000, lambda k, ten, q: verdict if k == stop_at else capi.FRAME_GO_ON)
    assert n == stop_at + 1
    ref = O.iterate(img, pts, tris, 0, ratio, RATE[0], stop_at + 1, literal=False)
    got = a.retrieve(capi.BUF_POINTS)
    used = np.zeros(pts.shape[0], bool); used[tris[:, :3].ravel()] = True
    assert np.array_equal(got[used].view(np.uint32), ref["points"][used].view(np.uint32))
    assert np.array_equal(got[~used], np.array([[0.1, 0.2], [ratio, -1.0]], np.float32))   # (clamped, not moved)
    if verdict == capi.FRAME_STOP_REPLAY:
        assert np.array_equal(a.retrieve(capi.BUF_TENERGY), ref["ten"])
        assert np.array_equal(a.retrieve(capi.BUF_COLNUM), ref["cn"])
        assert np.array_equal(a.retrieve(capi.BUF_COLACC)[:, :3], ref["ca"][:, :3])
        assert np.array_equal(a.retrieve(capi.BUF_GRADIENT)[used], ref["gr"][used])
    # ... and the descent goes on from there
    a.iterate(p, 7)
    ref2 = O.iterate(img, ref["points"], tris, 0, ratio, RATE[0], 7, literal=False)
    assert np.array_equal(a.retrieve(capi.BUF_TENERGY), ref2["ten"])
    assert np.array_equal(a.retrieve(capi.BUF_POINTS)[used].view(np.uint32), ref2["points"][used].view(np.uint32))
    a.close()


def test_short_runs_and_the_warp_flavour():
    """fewer frames than a persistent launch is worth (frame by frame on the two-kernel path), and the warp flavour against image B"""
    img, pts, tris, ratio = _mesh(300, 200)
    imgB = synth.displaced_raster(img, amp=5.0)
    colors = synth.mean_colors(img, pts, tris, ratio)
    a = capi.Context(0, 300, 200)
    a.set_image(capi.IMAGE_A, img); a.set_image(capi.IMAGE_B, imgB)
    a.upload(pts, tris, colors)
    p = capi.default_params(1)
    seen = []
    assert a.iterate_frames(p, 3, lambda k, ten, q: (seen.append(ten.copy()), capi.FRAME_GO_ON)[1]) == 3
    ref = O.iterate(imgB, pts, tris, 1, ratio, RATE[1], 3, colors=colors, literal=False)
    assert np.array_equal(seen[2], ref["ten"][: tris.shape[0]])
    assert np.array_equal(a.retrieve(capi.BUF_POINTS).view(np.uint32), ref["points"].view(np.uint32))
    n = a.iterate_frames(p, 40, lambda k, ten, q: capi.FRAME_STOP_REPLAY if k == 33 else capi.FRAME_GO_ON)
    assert n == 34
    ref2 = O.iterate(imgB, ref["points"], tris, 1, ratio, RATE[1], 34, colors=colors, literal=False)
    assert np.array_equal(a.retrieve(capi.BUF_TENERGY), ref2["ten"])
    assert np.array_equal(a.retrieve(capi.BUF_POINTS).view(np.uint32), ref2["points"].view(np.uint32))
    a.close()
