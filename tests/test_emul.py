"""The per-lane integer logic the HIP kernels run (tpose_amd/csrc/tp_raster.h: vertex stage, edge
walkers, row spans, prefix-sum lookups, finalize) compiled for the CPU and checked against the oracle.
This is a test harness around shared __host__ __device__ code -- not a product path."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from tpose_amd import synth
from util import case

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def em():
    os.makedirs(os.path.join(HERE, "_build"), exist_ok=True)
    so = os.path.join(HERE, "_build", "libtp_emul.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-o", so,
                           os.path.join(HERE, "emul", "emul.cpp")])
    lib = C.CDLL(so)
    lib.emul_reference_dp.restype = C.c_float
    return lib


def emul_moments(em, img, pts, tris, dp, ratio):
    NT = tris.shape[0]
    H, W = img.shape[:2]
    mom = np.zeros((13 * NT, 6), np.int64)
    em.emul_moments(img.ctypes.data_as(C.c_void_p), C.c_size_t(img.strides[0]), W, H,
                    pts.ctypes.data_as(C.c_void_p), tris.ctypes.data_as(C.c_void_p), NT,
                    C.c_float(dp), C.c_float(ratio), mom.ctypes.data_as(C.c_void_p), None)
    return mom


@pytest.mark.parametrize("W,H,grid", [(97, 61, (6, 4)), (300, 200, (15, 5)), (257, 131, (6, 4)),
                                      (640, 480, (50, 30)), (200, 150, None)])
@pytest.mark.parametrize("dp", [None, 0.2])
def test_span_walker_matches_oracle(em, W, H, grid, dp):
    img, _, pts, tris, ratio, colors = case(W, H, grid)
    NT = tris.shape[0]
    d = O.dp(0, NT) if dp is None else dp
    mom = emul_moments(em, img, pts, tris, d, ratio)
    assert np.array_equal(mom, O.moments(img, pts, tris, d, ratio))
    for fl in (0, 1):
        ten = np.zeros(13 * NT, np.int32); cn = np.zeros(13 * NT, np.int32); ca = np.zeros((13 * NT, 4), np.int32)
        em.emul_finalize(mom.ctypes.data_as(C.c_void_p), NT, fl, colors.ctypes.data_as(C.c_void_p),
                         ten.ctypes.data_as(C.c_void_p), cn.ctypes.data_as(C.c_void_p), ca.ctypes.data_as(C.c_void_p))
        rt, rc, rca, _ = O.finalize(mom, NT, fl, colors)
        assert np.array_equal(ten, rt) and np.array_equal(cn, rc)
        if fl == 0:
            assert np.array_equal(ca, rca)


def test_span_walker_on_triangle_soup(em):
    W, H = 200, 150
    img = synth.voronoi_raster(W, H, seed=3, sites=10)
    ratio = float(np.float32(W) / np.float32(H))
    rng = np.random.default_rng(1)
    for trial in range(30):
        NP = 30
        pts = (rng.random((NP, 2)).astype(np.float32) * 2 - 1) * np.float32(1.3)
        pts[:, 0] *= np.float32(ratio)
        if trial % 3 == 0:
            pts = (np.round(pts * 8) / 8).astype(np.float32)   # exact ties, horizontal/vertical edges
        if trial % 5 == 0:
            pts[:5] = pts[5:10]                                  # degenerate triangles
        tris = np.zeros((40, 4), np.int32)
        tris[:, :3] = rng.integers(0, NP, (40, 3))
        dp = [0.05, 0.0078125, 0.3][trial % 3]
        assert np.array_equal(emul_moments(em, img, pts, tris, dp, ratio), O.moments(img, pts, tris, dp, ratio)), trial


def test_reference_dp_matches_oracle(em):
    for fl in (0, 1):
        for NT in (2, 150, 3000, 12000, 40329):
            assert np.float32(em.emul_reference_dp(fl, NT)) == np.float32(O.dp(fl, NT))


def test_walker_floor_is_exact(em):
    """floor(x_r) == floor((N0 + r*step)/d) for 32 rows, including exact multiples, d = 1, the
    largest d, and the clamp region."""
    rng = np.random.default_rng(5)
    out = np.zeros(32, np.int32)
    em.emul_walker.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_int, C.c_void_p]
    cases = []
    for _ in range(20000):
        d = int(rng.choice([1, 2, 3, 255, 256, 257, (1 << 24) - 1, int(rng.integers(1, 1 << 24))]))
        step = int(rng.integers(-(1 << 24) + 1, 1 << 24))
        kind = rng.integers(0, 4)
        if kind == 0:
            N0 = int(rng.integers(-(1 << 40), 1 << 40))
        elif kind == 1:   # exact multiples of d: the floor sits on the knife edge
            N0 = d * int(rng.integers(-(1 << 15), 1 << 15))
            step = d * int(rng.integers(-3, 4)) if rng.integers(0, 2) else step
        elif kind == 2:   # just below / above a multiple
            N0 = d * int(rng.integers(-(1 << 15), 1 << 15)) + int(rng.choice([-1, 1]))
        else:
            N0 = int(rng.integers(-(1 << 30), 1 << 30)) * max(1, d // 7)
        if abs(N0) >= 1 << 41:
            continue
        cases.append((N0, step, d))
    for N0, step, d in cases:
        em.emul_walker(N0, step, d, 32, out.ctypes.data)
        for r in range(32):
            exact = (N0 + r * step) // d
            if abs(N0 // d) > (1 << 30):
                # clamped: only the side (far left / far right of any raster) matters
                assert (out[r] > (1 << 28)) == (exact > 0) and abs(int(out[r])) > (1 << 28)
            else:
                assert int(out[r]) == exact, (N0, step, d, r, int(out[r]), exact)


# ---- the table path: packed row prefix records, whole-line 24.40 walkers stepped like a lane of k_lines ----------
def emul_moments_table(em, img, pts, tris, dp, ratio, tl=8):
    NT, NP = tris.shape[0], pts.shape[0]
    H, W = img.shape[:2]
    mom = np.zeros((13 * NT, 6), np.int64)
    rc = em.emul_moments_table(img.ctypes.data_as(C.c_void_p), C.c_size_t(img.strides[0]), W, H,
                               pts.ctypes.data_as(C.c_void_p), tris.ctypes.data_as(C.c_void_p), NT, NP,
                               C.c_float(dp), C.c_float(ratio), tl, mom.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return mom


@pytest.mark.parametrize("W,H,grid", [(97, 61, (6, 4)), (300, 200, (15, 5)), (257, 131, (6, 4)), (200, 150, None),
                                      (640, 480, (50, 30))])
@pytest.mark.parametrize("dp", [None, 0.2])
@pytest.mark.parametrize("tl", [1, 8, 64])
def test_table_path_matches_oracle(em, W, H, grid, dp, tl):
    """moment(variant) = signed sum of three line sums, each one packed table record per row"""
    img, _, pts, tris, ratio, _ = case(W, H, grid)
    d = O.dp(0, tris.shape[0]) if dp is None else dp
    assert np.array_equal(emul_moments_table(em, img, pts, tris, d, ratio, tl), O.moments(img, pts, tris, d, ratio))


def test_table_path_on_triangle_soup(em):
    W, H = 200, 150
    img = synth.voronoi_raster(W, H, seed=3, sites=10)
    ratio = float(np.float32(W) / np.float32(H))
    rng = np.random.default_rng(4)
    for trial in range(40):
        NP = 30
        pts = (rng.random((NP, 2)).astype(np.float32) * 2 - 1) * np.float32(1.3)
        pts[:, 0] *= np.float32(ratio)
        if trial % 3 == 0:
            pts = (np.round(pts * 8) / 8).astype(np.float32)
        if trial % 5 == 0:
            pts[:5] = pts[5:10]
        if trial % 7 == 0:
            pts[10:14] *= np.float32(1e6)   # far outside: coordinates clamp
        if trial % 11 == 0:
            pts[3, 0] = np.float32("nan")
        tris = np.zeros((40, 4), np.int32)
        tris[:, :3] = rng.integers(0, NP, (40, 3))
        dp = [0.05, 0.0078125, 0.3][trial % 3]
        tl = [1, 4, 16][trial % 3]
        assert np.array_equal(emul_moments_table(em, img, pts, tris, dp, ratio, tl), O.moments(img, pts, tris, dp, ratio)), trial


@pytest.mark.parametrize("W,H", [(1, 1), (3, 2), (4, 3), (5, 2), (97, 5), (256, 2), (1011, 3)])
def test_prefix_records_are_exact(em, W, H):
    """every column 0..W of every row, from the packed 32-byte records (group prefix + masked pixel bytes)"""
    rng = np.random.default_rng(W * 31 + H)
    for kind in range(3):
        img = rng.integers(0, 256, (H, W, 4), dtype=np.uint8) if kind == 0 else \
            np.full((H, W, 4), 255, np.uint8) if kind == 1 else (rng.integers(0, 2, (H, W, 4), dtype=np.uint8) * 255)
        img = np.ascontiguousarray(img)
        assert em.emul_prefix_check(img.ctypes.data_as(C.c_void_p), C.c_size_t(img.strides[0]), W, H) == 0


def test_prefix_records_widest_row(em):
    """the packing limits: 16384 pixels of (255, 255, 255): 22-bit channel sums, 32-bit squares, 15-bit parity count"""
    W = 16384
    for val in ((255, 255, 255), (255, 254, 0), (1, 0, 0)):
        img = np.zeros((1, W, 4), np.uint8)
        img[..., :3] = val
        assert em.emul_prefix_check(img.ctypes.data_as(C.c_void_p), C.c_size_t(img.strides[0]), W, 1) == 0


def test_whole_line_walker_is_exact(em):
    """every row of a line, evaluated directly and stepped chunk-wise from ONE 24.40 set-up, equals exact integer arithmetic --
    up to the largest raster (16384 rows), the largest coordinates, knife-edge and lattice-aligned lines"""
    rng = np.random.default_rng(9)
    em.emul_line_check.argtypes = [C.c_int32] * 6 + [C.c_void_p]
    bad = C.c_int32(0)
    lo, hi = -(1 << 22), 1 << 23
    n = 0
    for trial in range(3000):
        kind = trial % 6
        H = [16384, 2048, 4096, 600, 16384, 97][kind]
        if kind == 0:    # anything in the coordinate range, tallest raster
            Xa, Ya, Xb, Yb = (int(rng.integers(lo, hi + 1)) for _ in range(4))
        elif kind == 1:  # raster-sized
            Xa, Xb = (int(rng.integers(-100 * 256, 2148 * 256)) for _ in range(2))
            Ya, Yb = (int(rng.integers(-100 * 256, 2148 * 256)) for _ in range(2))
        elif kind == 2:  # lattice aligned: exact ties everywhere
            Xa, Ya, Xb, Yb = (256 * int(rng.integers(-50, 4200)) + 128 for _ in range(4))
        elif kind == 3:  # nearly horizontal / nearly vertical
            Xa, Ya = int(rng.integers(0, 600 * 256)), int(rng.integers(0, 600 * 256))
            Xb, Yb = (Xa + int(rng.integers(-5, 6)), int(rng.integers(0, 600 * 256))) if trial % 2 else \
                     (int(rng.integers(lo, hi)), Ya + int(rng.integers(1, 700)))
        elif kind == 4:  # extremes
            Xa, Ya, Xb, Yb = int(rng.choice([lo, hi])), lo, int(rng.choice([lo, hi])), hi
        else:
            Xa, Ya, Xb, Yb = (int(rng.integers(-30 * 256, 130 * 256)) for _ in range(4))
        for tl in (1, 16):
            r = em.emul_line_check(Xa, Ya, Xb, Yb, H, tl, C.byref(bad))
            assert r == 0, (Xa, Ya, Xb, Yb, H, tl, bad.value)
            n += 1
    assert n == 6000


# ------------------------------------------------------------------------------------------------
# Persistent grad-iter kernel (tpose_amd/csrc/tp_plan.h + tp_persist.h): the plan and the lane functions replayed on
# the CPU, workgroup by workgroup, with the position mailbox as a plain array -- everything but the device-side waiting.
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def emp():
    os.makedirs(os.path.join(HERE, "_build"), exist_ok=True)
    so = os.path.join(HERE, "_build", "libtp_emul_persist.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-o", so,
                           os.path.join(HERE, "emul", "emul_persist.cpp")])
    return C.CDLL(so)


def emul_persist(emp, sweep, pts, tris, flavour, ratio, rate, iters, colors=None, dp=None, max_parts=256, lds=160 * 1024):
    NT, NP = tris.shape[0], pts.shape[0]
    H, W = sweep.shape[:2]
    d = O.dp(flavour, NT) if dp is None else dp
    p = np.ascontiguousarray(pts.copy())
    ca = np.ascontiguousarray(np.tile(colors, (13, 1)).astype(np.int32)) if colors is not None else None
    stats = np.zeros(16, np.int64)
    out = dict(ten=np.zeros(13 * NT, np.int32), cn=np.zeros(13 * NT, np.int32), ca=np.zeros((13 * NT, 4), np.int32), gr=np.zeros((NP, 2), np.int32))
    rc = emp.emul_persist(sweep.ctypes.data_as(C.c_void_p), C.c_size_t(sweep.strides[0]), W, H, p.ctypes.data_as(C.c_void_p), NP,
                          tris.ctypes.data_as(C.c_void_p), NT, ca.ctypes.data_as(C.c_void_p) if ca is not None else None,
                          flavour, C.c_float(d), C.c_float(ratio), C.c_float(rate), iters, max_parts, lds,
                          stats.ctypes.data_as(C.c_void_p), out["ten"].ctypes.data_as(C.c_void_p), out["cn"].ctypes.data_as(C.c_void_p),
                          out["ca"].ctypes.data_as(C.c_void_p), out["gr"].ctypes.data_as(C.c_void_p))
    stats_out.clear(); stats_out.update(out)
    return rc, p, stats


stats_out = {}


@pytest.mark.parametrize("W,H,grid", [(64, 48, (6, 4)), (300, 200, (15, 5)), (257, 131, (6, 4)), (128, 32, None)])
@pytest.mark.parametrize("flavour", [0, 1])
@pytest.mark.parametrize("max_parts", [1, 5, 256])
def test_persistent_plan_and_lanes_match_oracle(emp, W, H, grid, flavour, max_parts):
    img, imgB, pts, tris, ratio, colors = case(W, H, grid)
    sweep = imgB if flavour else img
    rate = 0.00003 if flavour else 0.00005
    rc, p, stats = emul_persist(emp, sweep, pts, tris, flavour, ratio, rate, 6, colors=colors if flavour else None,
                                max_parts=max_parts, lds=1 << 20)
    assert rc == 0, (rc, stats)
    ref = O.iterate(sweep, pts, tris, flavour, ratio, rate, 6, colors=colors if flavour else None, literal=False)
    assert np.array_equal(p.view(np.uint32), ref["points"].view(np.uint32))
    assert 1 <= stats[1] <= max_parts
    # the last grad-iter's outputs in the reference's layout: all 13 variants of every triangle, and the gradient
    assert np.array_equal(stats_out["ten"], ref["ten"]) and np.array_equal(stats_out["cn"], ref["cn"])
    assert np.array_equal(stats_out["gr"], ref["gr"].reshape(-1, 2))
    if flavour == 0:
        assert np.array_equal(stats_out["ca"], ref["ca"].reshape(-1, 4))


def test_persistent_record_cache_over_many_grad_iters(emp):
    """the records a lane keeps from one grad-iter to the next: fast-moving vertices (stale records, lines that outgrow
    the rows a lane keeps) over 60 grad-iters"""
    W, H, grid = 300, 200, (15, 5)
    img, _, pts, tris, ratio, _ = case(W, H, grid)
    seen = set()
    for rate, max_parts in ((0.00005, 7), (0.0004, 7), (0.0004, 12)):
        rc, p, stats = emul_persist(emp, img, pts, tris, 0, ratio, rate, 60, max_parts=max_parts)
        assert rc == 0
        seen.add(int(stats[13]))
        ref = O.iterate(img, pts, tris, 0, ratio, rate, 60, literal=False)
        assert np.array_equal(p.view(np.uint32), ref["points"].view(np.uint32)), rate
    # (the replay of the rows beyond the registers: eight rows per lane in LDS -- both words of a slot's crossing columns -- with seven patches)
    assert 8 in seen, seen


def test_persistent_lines_are_cut_again_on_the_device(emp):
    """every PK_RECUT grad-iters a patch counts the chunks of its lines again from the positions it has: 200 grad-iters of a
    fast descent change the cut several times and never the result; the outputs of the last grad-iter come from base lines
    cut long before"""
    W, H, grid = 300, 200, (15, 5)
    img, _, pts, tris, ratio, _ = case(W, H, grid)
    recuts = np.zeros(1, np.int64)
    emp.emul_persist_walk_stats(None, 0, recuts.ctypes.data_as(C.c_void_p))
    try:
        rc, p, stats = emul_persist(emp, img, pts, tris, 0, ratio, 0.0004, 200, max_parts=7)
    finally:
        emp.emul_persist_walk_stats(None, 0, None)
    assert rc == 0
    assert recuts[0] > stats[1], "no patch cut its lines again after the first grad-iter"
    ref = O.iterate(img, pts, tris, 0, ratio, 0.0004, 200, literal=False)
    assert np.array_equal(p.view(np.uint32), ref["points"].view(np.uint32))
    assert np.array_equal(stats_out["ten"], ref["ten"]) and np.array_equal(stats_out["gr"], ref["gr"])
    assert emp.emul_magic_check() == 0
    assert emp.emul_fold_check() == 0   # the packed line sums at a 4096 x 4096 all-white raster


def test_persistent_plan_on_soups_and_bad_vertices(emp):
    """triangle soups (edges shared by many triangles, duplicated triangles, vertices outside the domain, NaN): the plan
    either refuses (an edge naming one vertex twice) or replays to the oracle's bits"""
    W, H = 200, 150
    img = synth.voronoi_raster(W, H, seed=3, sites=10)
    ratio = float(np.float32(W) / np.float32(H))
    rng = np.random.default_rng(5)
    ran = 0
    for trial in range(24):
        NP = 30
        pts = (rng.random((NP, 2)).astype(np.float32) * 2 - 1) * np.float32(1.3)
        pts[:, 0] *= np.float32(ratio)
        if trial % 5 == 0:
            pts[10:14] *= np.float32(1e6)
        if trial % 7 == 0:
            pts[3, 0] = np.float32("nan")
        tris = np.zeros((40, 4), np.int32)
        tris[:, :3] = rng.integers(0, NP, (40, 3))
        if trial % 2 == 0:   # no degenerate triangles: the plan must accept
            for t in range(40):
                while len(set(tris[t, :3].tolist())) < 3:
                    tris[t, :3] = rng.integers(0, NP, 3)
        rc, p, stats = emul_persist(emp, img, pts, tris, 0, ratio, 0.00005, 4, dp=[0.05, 0.3][trial % 2], max_parts=1 + trial % 6, lds=1 << 20)
        if trial % 2 == 0:
            assert rc == 0
        if rc == 0:
            ran += 1
            ref = O.iterate(img, pts, tris, 0, ratio, 0.00005, 4, dp_=[0.05, 0.3][trial % 2], literal=False)
            used = np.zeros(NP, bool)
            used[tris[:, :3].ravel()] = True   # (vertices no triangle uses are left to the last grad-iter's k_update, which clamps them)
            assert np.array_equal(p.view(np.uint32)[used], ref["points"].view(np.uint32)[used]), trial
            assert np.array_equal(p.view(np.uint32)[~used], pts.view(np.uint32)[~used])
        else:
            assert rc == -1 and stats[0] == 0
    assert ran >= 12


def test_persistent_plan_statistics(emp):
    """the cut at the metric size: every CU gets a patch, patches are balanced, few lines are walked twice"""
    W = H = 2048
    img, pts, tris, he, ratio = synth.workload(W, H, 3000)
    NP, NT = pts.shape[0], tris.shape[0]
    stats = np.zeros(16, np.int64)
    owner = np.zeros(NP, np.int32)
    rc = emp.emul_plan(pts.ctypes.data_as(C.c_void_p), NP, tris.ctypes.data_as(C.c_void_p), NT, W, H, C.c_float(ratio),
                       C.c_float(10.0), 256, 160 * 1024, owner.ctypes.data_as(C.c_void_p), None, stats.ctypes.data_as(C.c_void_p))
    assert rc == 0 and stats[0] == 1 and stats[1] == 256
    assert stats[3] <= 1.15 * 9 * stats[5]          # lines walked vs 9 per edge
    assert stats[6] <= 1.3 * stats[7]               # heaviest patch vs the mean
    # LDS per workgroup: the tables alone when no patch takes more than 16 rows per lane (this mesh: stats[12] = rows per lane of the largest patch), and
    # on top, when the plan keeps rows beyond the registers in LDS: four rows (plans of up to 18 rows per lane) or eight of 768 records of 16 bytes,
    # plus 16 bytes of crossing columns per slot (tp_plan.h: PK_LDS_ROW_BYTES, PK_LDS_COL_BYTES)
    lds_rows = 0 if stats[12] <= 16 else 4 if stats[12] <= 18 else 8
    assert stats[2] <= 64 * 1024 + (lds_rows * 768 * 16 + 768 * 16 if lds_rows else 0)
    assert (np.bincount(owner[owner >= 0], minlength=256) > 0).all()


@pytest.mark.parametrize("W,H", [(1, 1), (2, 1), (3, 2), (5, 2), (97, 5), (256, 2), (1011, 3), (4096, 3)])
def test_pixel_records_are_exact(em, W, H):
    """every column 0..W of every row from the 16-byte pixel records, and sums of up to 16 of them unpacked once -- at the
    packing limits too (4096 pixels of 255s: 20-bit channel sums, 30-bit squares, 13-bit parity count, 16 records added)"""
    rng = np.random.default_rng(W * 31 + H)
    for kind in range(3):
        img = rng.integers(0, 256, (H, W, 4), dtype=np.uint8) if kind == 0 else \
            np.full((H, W, 4), 255, np.uint8) if kind == 1 else (rng.integers(0, 2, (H, W, 4), dtype=np.uint8) * 255)
        img = np.ascontiguousarray(img)
        assert em.emul_px_check(img.ctypes.data_as(C.c_void_p), C.c_size_t(img.strides[0]), W, H) == 0


def test_persistent_patch_without_vertices(emp):
    """a crowded mesh (3000 triangles squeezed into ~200 x 100 pixels of a 2048^2 raster, the four corners left in place) is cut
    into patches of which one owns nothing: its threads must decide so at the first cut of a launch instead of walking whatever
    their registers held (the emulator fills the lane caches with garbage like a kernel's registers at launch)"""
    W = H = 2048
    img, pts, tris, he, ratio = synth.workload(W, H, 3000)
    bad = pts.copy()
    bad[4:] = bad[4:] * np.float32(0.1) * np.array([1.0, 0.5], np.float32) + np.float32(0.3)
    rc, p, stats = emul_persist(emp, img, bad, tris, 0, ratio, 0.00005, 2, max_parts=256, lds=160 * 1024 - 512)
    assert rc == 0
    ref = O.iterate(img, bad, tris, 0, ratio, 0.00005, 2, literal=False)
    assert np.array_equal(p.view(np.uint32), ref["points"].view(np.uint32))
    assert np.array_equal(stats_out["ten"], ref["ten"])


def test_packed_variant_moments_and_energy_equal_the_general_form(emp):
    """round 5: P6 sums the line sums as packed words and takes the energy in 24/32-bit arithmetic (tp_persist.h: pk_signed_packed,
    pk_energy_var); the general 64-bit form is the reference here -- random and extreme fields, both flavours, colours outside a byte"""
    emp.emul_packed_check.argtypes = [C.c_uint64, C.c_int]
    assert emp.emul_packed_check(12345, 200000) == 0
