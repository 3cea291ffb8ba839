"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle's literal
reference form after an independent brute force agreed): the oracle on CPU, the HIP path on the GPU."""
import glob
import os

import numpy as np
import pytest

from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))


def load(path):
    z = np.load(path)
    return {k: z[k] for k in z.files}


def test_fixtures_exist():
    assert len(FILES) >= 5


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_oracle_reproduces_golden(path):
    g = load(path)
    fl = int(g["flavour"])
    col = g["colors"] if fl else None
    for literal in (True, False):
        st = O.iterate(g["image"], g["points"], g["triangles"], fl, float(g["ratio"]), float(g["rate"]), 1,
                       colors=col, literal=literal)
        assert np.array_equal(st["cn"], g["cn1"]) and np.array_equal(st["ten"], g["ten1"])
        assert np.array_equal(st["gr"], g["gr1"])
        if fl == 0:
            assert np.array_equal(st["ca"], g["ca1"])
    for key in g:
        if key.startswith("points_"):
            n = int(key.split("_")[1])
            st = O.iterate(g["image"], g["points"], g["triangles"], fl, float(g["ratio"]), float(g["rate"]), n,
                           colors=col, literal=False)
            assert np.array_equal(st["points"].view(np.uint32), g[key].view(np.uint32))
            assert np.array_equal(st["ten"], g["ten_%d" % n])


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_hip_reproduces_golden(path):
    from tpose_amd import capi
    g = load(path)
    fl = int(g["flavour"])
    W, H = int(g["W"]), int(g["H"])
    slot = capi.IMAGE_B if fl else capi.IMAGE_A
    ctx = capi.Context(0, W, H)
    ctx.set_ratio(float(g["ratio"]))
    ctx.set_image(slot, g["image"])
    for key in sorted(k for k in g if k.startswith("points_")):
        n = int(key.split("_")[1])
        ctx.upload(g["points"], g["triangles"], g["colors"] if fl else None)
        if n == 1:  # piecewise API: every buffer of the first frame
            ctx.accumulate(fl, slot)
            ctx.energy(fl)
            assert np.array_equal(ctx.retrieve(capi.BUF_COLNUM), g["cn1"])
            assert np.array_equal(ctx.retrieve(capi.BUF_TENERGY), g["ten1"])
            assert np.array_equal(ctx.retrieve(capi.BUF_COLACC), g["ca1"])
            ctx.shift(float(g["rate"]))
            assert np.array_equal(ctx.retrieve(capi.BUF_GRADIENT), g["gr1"])
        else:
            ctx.iterate(capi.default_params(fl, image_slot=slot), n)
            assert np.array_equal(ctx.retrieve(capi.BUF_TENERGY), g["ten_%d" % n])
        assert np.array_equal(ctx.retrieve(capi.BUF_POINTS).view(np.uint32), g[key].view(np.uint32))
    ctx.close()
