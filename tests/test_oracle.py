"""CPU tests of the oracle: against an independent NumPy brute force, against itself (literal
two-pass form == single-sweep moment form), and against the known answers SURVEY.md section 8c lists
for the host half (geterr/maxerrid).  The reference holds no golden vectors for the GPU half."""
import numpy as np
import pytest

import brute
from oracle import oracle as O
from tpose_amd import synth
from util import RATE, case


def test_u8_texture_roundtrip_is_exact():
    v = np.arange(256, dtype=np.float32)
    t = (v / np.float32(255)).astype(np.float32)
    assert np.array_equal((np.float32(255) * t).astype(np.float32), v)


def test_dp_law():
    for fl in (0, 1):
        for NT in (2, 150, 1000, 3000, 12000):
            assert np.float32(O.dp(fl, NT)) == brute.dp_law(fl, NT)
    assert abs(O.dp(0, 3000) - 0.01) < 1e-9 and abs(O.dp(1, 1000) - 0.005) < 1e-9


@pytest.mark.parametrize("W,H,grid", [(64, 48, (6, 4)), (97, 61, (6, 4)), (61, 97, (6, 4)), (33, 17, None)])
@pytest.mark.parametrize("flavour", [0, 1])
def test_literal_matches_bruteforce(W, H, grid, flavour):
    img, imgB, pts, tris, ratio, colors = case(W, H, grid)
    sweep = imgB if flavour else img
    dp = O.dp(flavour, tris.shape[0])
    if flavour == 0:
        cn, ca = O.accumulate_literal(sweep, pts, tris, dp, ratio)
    else:
        ca = np.tile(colors, (13, 1)).astype(np.int32)
        cn, _ = O.accumulate_literal(sweep, pts, tris, dp, ratio, count_only=True, ca=ca)
    ten = O.energy_literal(sweep, pts, tris, dp, ratio, flavour, cn, ca)
    bcn, bca, bten = brute.evaluate(sweep, pts, tris, flavour, ratio, colors=colors)
    assert np.array_equal(cn, bcn) and np.array_equal(ca, bca) and np.array_equal(ten, bten)


def test_vertex_stage_and_coverage_match_bruteforce():
    W, H = 50, 40
    img, _, pts, tris, ratio, _ = case(W, H, (6, 4))
    dp = O.dp(0, tris.shape[0])
    for t in (0, 7, 23):
        for i in range(13):
            xy = O.variant_vertices(pts, tris, t, i, dp, ratio, W, H)
            assert list(xy) == brute.variant_xy(pts, tris[t], i, dp, ratio, W, H)
            m = brute.coverage_mask(xy, W, H)
            got = np.array([[O.covered(xy, c, r) for c in range(W)] for r in range(H)])
            assert np.array_equal(m, got)


@pytest.mark.parametrize("flavour", [0, 1])
def test_moment_form_equals_literal_form(flavour):
    img, imgB, pts, tris, ratio, colors = case(300, 200, (15, 5))
    sweep = imgB if flavour else img
    a = O.iterate(sweep, pts, tris, flavour, ratio, RATE[flavour], 5, colors=colors if flavour else None, literal=True)
    b = O.iterate(sweep, pts, tris, flavour, ratio, RATE[flavour], 5, colors=colors if flavour else None, literal=False)
    for k in ("ten", "cn", "gr"):
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(a["points"].view(np.uint32), b["points"].view(np.uint32))
    assert not np.array_equal(a["points"], pts)  # something moved


def test_base_variants_tile_the_raster_exactly():
    img, _, pts, tris, ratio, _ = case(257, 131, (6, 4))
    NT = tris.shape[0]
    mom = O.moments(img, pts, tris, O.dp(0, NT), ratio)
    assert mom[:NT, 0].sum() == 257 * 131
    assert np.array_equal(mom[:NT, 2:5].sum(axis=0), img[:, :, :3].astype(np.int64).sum(axis=(0, 1)))


def test_gradient_and_shift_semantics():
    # hand-made energies: gradient.cs indexing and shift.cs clamping
    tris = np.array([[0, 4, 5, 0], [4, 5, 1, 0]], np.int32)
    NT, NP = 2, 6
    ten = np.arange(13 * NT, dtype=np.int32) ** 2
    gr = O.gradient(ten, tris, NP)
    exp = np.zeros((NP, 2), np.int64)
    for t in range(NT):
        for s in range(3):
            v = tris[t, s]
            exp[v, 0] += int(ten[(4 * s + 1) * NT + t]) - int(ten[(4 * s + 2) * NT + t])
            exp[v, 1] += int(ten[(4 * s + 3) * NT + t]) - int(ten[(4 * s + 4) * NT + t])
    assert np.array_equal(gr, exp)
    pts = np.array([[-1.5, -1], [-1.5, 1], [1.5, -1], [1.5, 1], [1.6, 0.25], [0.5, -1.0]], np.float32)
    g = np.array([[9, 9]] * 4 + [[65536 * 1000, -65536 * 2000], [65536 * 4000, 65536 * 4000]], np.int32)
    out = O.shift(pts, g, 1.5, 0.00005)
    assert np.array_equal(out[:4], pts[:4])                       # corners never move
    assert out[4, 0] == np.float32(1.5)                           # clamped, x-gradient zeroed
    assert out[4, 1] == np.float32(0.25) - np.float32(0.00005) * np.float32(-65536 * 2000) / 256 / 256
    assert out[5, 1] == np.float32(-1.0)                          # on the edge: sticks
    assert out[5, 0] == np.float32(0.5) - np.float32(np.float32(0.00005) * np.float32(65536 * 4000)) / 256 / 256


def test_geterr_known_answers():
    """SURVEY.md section 8c fixture (vi): terr={100,400} from toterr=1 -> 499 then 0; maxerr 20; maxerrid 1."""
    st = O.ErrState()
    terr = np.array([100, 400], np.int32)
    assert st.geterr(terr, 2) == 499.0
    assert st.geterr(terr, 2) == 0.0
    assert float(st.st[3]) == 20.0
    assert st.maxerrid(terr, 2) == 1
    assert st.gettoterr(terr, 2) == 500.0
    assert st.maxerrid(np.zeros(2, np.int32), 2) == -1
