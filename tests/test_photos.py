"""The reference's own pictures under the hot path (round 6).  BASELINE.json's configs 1, 2, 3 and 5 name resource/fruit.png, meninas.png,
imageA/B.png and shoeA/B.png; tests/golden/photos/ holds their decoded RGB bytes (tests/golden/make_photos.py, run in the build container --
data only, the PNGs do not travel).  Here: the fixtures are what they say they are, the integer resampler is deterministic, and the HIP
path is the oracle's, bit for bit, on them -- config 1 as written on fruit.png's window, and the metric size (2048^2 / 3000 triangles) on
meninas.png resampled to 2048 x 2048, fresh and after a thousand grad-iters of re-plans weighted by the vertices' measured speeds.
Configs 2, 3 and 5 on the pictures: tests/test_configs.py."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from tpose_amd import capi, photos, synth
from util import RATE

REF = "/root/reference/resource"


def test_fixtures_decode_to_the_bytes_their_index_names():
    idx = photos.index()
    assert set(photos.NAMES) <= set(idx)
    for n in photos.NAMES:
        img = photos.load(n)           # (verifies the sha256 of the RGB bytes)
        assert img.dtype == np.uint8 and img.shape == (idx[n]["h"], idx[n]["w"], 4) and int(img[:, :, 3].min()) == 255
    assert photos.load("fruit").shape[:2] == (674, 1011) and photos.load("meninas").shape[:2] == (1381, 1200)


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists in the build container only")
def test_fixtures_are_the_reference_pictures():
    Image = pytest.importorskip("PIL.Image")
    for n in photos.NAMES:
        ref = np.asarray(Image.open(os.path.join(REF, n + ".png")).convert("RGBA"))
        assert np.array_equal(photos.load(n), ref), n


def test_integer_resampler():
    img = photos.load("shoeA")
    H, W = img.shape[:2]
    assert np.array_equal(photos.resample_int(img, W, H), img)                       # same size: the same bytes
    flat = np.full((7, 5, 4), 93, np.uint8)
    assert np.all(photos.resample_int(flat, 64, 48) == 93)                            # weights sum to 2^16
    ramp = np.zeros((1, 2, 4), np.uint8); ramp[0, 1] = 200
    out = photos.resample_int(ramp, 4, 1)[0, :, 0].tolist()
    assert out == [0, 50, 150, 200]                                                   # centres at 0.25, 0.75, 1.25, 1.75 source pixels (clamped ends)
    a = photos.resample_int(img, 640, 360); b = photos.resample_int(img, 640, 360)
    assert np.array_equal(a, b) and a.shape == (360, 640, 4)
    assert photos.window("fruit").shape == (449, 674, 4) and photos.window("meninas").shape == (920, 800, 4)   # main.cpp:53: image / 1.5, truncated


def test_oracle_forms_agree_on_fruit():
    """the literal two-pass form and the moment form of the oracle, config 1's raster and mesh, 3 grad-iters"""
    img = photos.window("fruit")
    H, W = img.shape[:2]
    ratio = float(np.float32(W) / np.float32(H))
    pts, tris, _ = synth.grid_triangulation(15, 5, ratio=ratio)
    a = O.iterate(img, pts, tris, 0, ratio, RATE[0], 3, literal=True)
    b = O.iterate(img, pts, tris, 0, ratio, RATE[0], 3, literal=False)
    for k in ("ten", "cn", "ca", "gr"):
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(a["points"].view(np.uint32), b["points"].view(np.uint32))
    assert int(a["cn"][: tris.shape[0]].sum()) == W * H     # the base variants tile the window exactly once


def _compare(ctx, ref, tag=""):
    assert np.array_equal(ctx.retrieve(capi.BUF_TENERGY), ref["ten"]), "tenergy " + tag
    assert np.array_equal(ctx.retrieve(capi.BUF_COLNUM), ref["cn"]), "colnum " + tag
    assert np.array_equal(ctx.retrieve(capi.BUF_COLACC)[:, :3], ref["ca"][:, :3]), "colacc " + tag
    assert np.array_equal(ctx.retrieve(capi.BUF_GRADIENT), ref["gr"]), "gradient " + tag
    assert np.array_equal(ctx.retrieve(capi.BUF_POINTS).view(np.uint32), ref["points"].view(np.uint32)), "points " + tag


@pytest.mark.gpu
def test_config1_on_fruit():
    """BASELINE config 1 on the picture it names: fruit.png at the reference's window (1011 x 674 / 1.5 = 674 x 449), 150 triangles,
    200 grad-iters, triangulate flavour (software/triangulate/main.cpp:53, 190-204) -- in one call, and in calls of 37"""
    img = photos.window("fruit")
    H, W = img.shape[:2]
    ratio = float(np.float32(W) / np.float32(H))
    pts, tris, _ = synth.grid_triangulation(15, 5, ratio=ratio)
    ctx = capi.Context(0, W, H)
    ctx.set_image(capi.IMAGE_A, img)
    ctx.upload(pts, tris)
    p = capi.default_params(0)
    ctx.iterate(p, 200)
    ctx.synchronize()
    assert ctx.info(capi.INFO_PERSIST_ITERS) == 200 and ctx.info(capi.INFO_PERSIST_FAILURES) == 0
    ref = O.iterate(img, pts, tris, 0, ratio, RATE[0], 200, literal=False)
    _compare(ctx, ref)
    ctx.upload(pts, tris)
    done = 0
    while done < 200:
        k = min(37, 200 - done)
        ctx.iterate(p, k)
        done += k
    _compare(ctx, ref, "calls of 37")
    ctx.close()


@pytest.mark.gpu
def test_the_metric_size_on_meninas():
    """2048^2 / 3000 triangles on resource/meninas.png resampled to 2048 x 2048 (config 2's picture at the metric's size; bench.py's
    `on_reference_photo` figures): 12 grad-iters in one persistent launch and 5 more in a warm one; then 1100 more on the device -- vertices
    jump a pixel or more per grad-iter on this picture, the planner re-weighs the patches by the speeds the kernel measures (tp_get_info 14)
    -- and 16 grad-iters from that state, all under the oracle, 0 ulp"""
    W = H = 2048
    img = photos.resample_int(photos.load("meninas"), W, H)
    _, pts, tris, he, ratio = synth.workload(W, H, 3000)
    ctx = capi.Context(0, W, H)
    ctx.set_image(capi.IMAGE_A, img)
    ctx.upload(pts, tris)
    p = capi.default_params(0)
    ctx.iterate(p, 12)
    ref = O.iterate(img, pts, tris, 0, ratio, RATE[0], 12, literal=False)
    _compare(ctx, ref, "12 grad-iters")
    ctx.iterate(p, 5)
    ref = O.iterate(img, ref["points"], tris, 0, ratio, RATE[0], 5, literal=False)
    _compare(ctx, ref, "5 more")
    ctx.iterate(p, 1100)
    old = ctx.retrieve(capi.BUF_POINTS)
    ctx.iterate(p, 9)
    ctx.iterate(p, 7)
    ref = O.iterate(img, old, tris, 0, ratio, RATE[0], 16, literal=False)
    _compare(ctx, ref, "16 grad-iters behind 1117")
    assert ctx.info(capi.INFO_PERSIST_ITERS) == 12 + 5 + 1100 + 16 and ctx.info(capi.INFO_PERSIST_FAILURES) == 0
    assert ctx.info(capi.INFO_REPLANS) >= 1
    ctx.close()
