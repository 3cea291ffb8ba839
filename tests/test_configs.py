"""BASELINE.json configs 2, 3 and 5 at their workloads -- on the reference's OWN pictures (round 6: resource/meninas.png, imageA/B.png,
shoeA/B.png, decoded into tests/golden/photos/ by tests/golden/make_photos.py; the PNGs themselves do not travel) and on synthetic
rasters of the same sizes: through the headless harnesses -- the reference's frame schedules
(software/triangulate/main.cpp:206-351, software/warp/main.cpp:214-283,
tests/compute_fundamental_mat/main.cpp:137-184) over the C++ host mirror and the HIP C ABI."""
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from tpose_amd import capi, photos, synth
from test_harness import HOST, build_cpu, build_gpu, records, run, write_ppm

LADDER = [50, 100, 200, 300, 400, 500, 600, 700, 800, 900, 1000, 1500, 2000, 2500, 3000]


def photo_like(W, H, seed, sites):
    """Voronoi + noise raster with photograph-like contrast between neighbouring regions: the reference's fixed-step
    descent (rate * dE / 65536, shift.cs:45) only meets its convergence test (relative change < 1e-4) on such rasters"""
    img = synth.voronoi_raster(W, H, seed=seed, sites=sites)
    rgb = img[:, :, :3].astype(np.float32)
    img[:, :, :3] = np.clip(128.0 + (rgb - 128.0) * 0.1 + 0.5, 0, 255).astype(np.uint8)
    return img


def config2_picture(kind):
    """the 1200 x 1381 picture config 2 names (resource/meninas.png), or its synthetic stand-in of rounds 1-5"""
    return photos.load("meninas") if kind == "meninas" else photo_like(1200, 1381, 1234, 160)


def read_level(path, level):
    """(ratio, triangles int32[NT,4], halfedges int32[3NT], colors int32[NT,4], points f32[NP,2], origin f32[NP,2])"""
    data, off = open(path, "rb").read(), 0
    for k in range(level + 1):
        ratio = float(np.frombuffer(data, np.float32, 1, off)[0])
        NT = int(np.frombuffer(data, np.int32, 1, off + 4)[0])
        body = np.frombuffer(data, np.int32, 9 * NT, off + 8).reshape(NT, 9)
        NP = int(np.frombuffer(data, np.int32, 1, off + 8 + 36 * NT)[0])
        pts = np.frombuffer(data, np.float32, 4 * NP, off + 12 + 36 * NT).reshape(NP, 4)
        off += 12 + 36 * NT + 16 * NP
    tris = np.zeros((NT, 4), np.int32); tris[:, :3] = body[:, 0:3]
    cols = np.ones((NT, 4), np.int32); cols[:, :3] = body[:, 6:9]
    return ratio, tris, body[:, 3:6].ravel().copy(), cols, pts[:, 0:2].copy(), pts[:, 2:4].copy()


def check_halfedges(tris, he):
    """twin(twin(h)) == h, twins run the same edge the other way, an edge without a twin is unique"""
    NT = tris.shape[0]
    org = tris[:, :3].ravel()
    dst = tris[:, [1, 2, 0]].ravel()
    h = np.arange(3 * NT)
    has = he >= 0
    assert np.all(he[has] < 3 * NT)
    assert np.array_equal(he[he[has]], h[has])
    assert np.array_equal(org[he[has]], dst[has]) and np.array_equal(dst[he[has]], org[has])
    key = org.astype(np.int64) * (1 << 32) + dst
    assert np.unique(key).size == key.size  # no directed edge twice


@pytest.mark.gpu
def test_config2_on_meninas_itself(tmp_path):
    """config 2 on the picture it names.  On resource/meninas.png the reference's convergence test (relative change of the summed energy below
    1e-4, software/triangulate/main.cpp:210) needs ~10 000 frames per split at the coarse levels -- 300 000 frames reach 45 triangles
    (profiles/r06_config2_meninas.txt) -- so the suite runs the schedule's first 120 000 frames on the HIP path: levels exported at 6, 10 and 14
    triangles, consistent half-edges, and the state of the last level one more grad-iter on, bit-equal to the oracle ON THE PICTURE"""
    ppm = str(tmp_path / "meninas.ppm")
    write_ppm(ppm, photos.load("meninas"))
    gpu = build_gpu("triangulate")
    tri = str(tmp_path / "m.tri")
    out = run(gpu, "-i", ppm, "-o", tri, "-levels", "6,10,14", "-window", "1.5", "-maxframes", "120000", "-quiet")
    recs = records(tri)
    assert "levels written" in out and len(recs) >= 2, out
    for (ratio, NT, NP), want in zip(recs, [6, 10, 14]):
        assert want <= NT <= want + 2 and abs(ratio - 1200 / 1381) < 1e-6
    ratio, tris, he, cols, pts, org = read_level(tri, len(recs) - 1)
    check_halfedges(tris, he)
    assert np.array_equal(pts, org)
    # the raster the harness swept: its own GL_LINEAR-style resampling of the picture (image_io.hpp) -- here the integer one, so the
    # comparison below is HIP against oracle on THIS raster, from the exported state
    img = photos.window("meninas")
    ctx = capi.Context(0, 800, 920)
    ctx.set_ratio(ratio)
    ctx.set_image(capi.IMAGE_A, img)
    ctx.upload(pts, tris)
    ctx.iterate(capi.default_params(0), 6)
    ref = O.iterate(img, pts, tris, 0, ratio, 0.00005, 6, literal=False)
    assert np.array_equal(ctx.retrieve(capi.BUF_TENERGY), ref["ten"])
    assert np.array_equal(ctx.retrieve(capi.BUF_COLNUM), ref["cn"])
    assert np.array_equal(ctx.retrieve(capi.BUF_POINTS).view(np.uint32), ref["points"].view(np.uint32))
    ctx.close()


@pytest.mark.gpu
def test_config2_full_topology_schedule_to_3000_triangles(tmp_path, kind="synthetic"):
    """config 2: the window the reference opens for resource/meninas.png (1200x1381 / 1.5 = 800x920), the whole
    schedule -- flip sets with flip-back, splits at the worst triangle, prune / wide-angle flips / collapses every
    frame, four readbacks per frame -- from 2 to 3000 triangles on the HIP path, on the picture's synthetic stand-in (on the picture
    itself the ladder is millions of frames long: test_config2_on_meninas_itself)"""
    W, H = 1200, 1381
    ppm = str(tmp_path / "meninas_like.ppm")
    write_ppm(ppm, config2_picture(kind))
    gpu = build_gpu("triangulate")
    tri = str(tmp_path / "c2.tri")
    out = run(gpu, "-i", ppm, "-o", tri, "-levels", ",".join(str(v) for v in LADDER), "-window", "1.5", "-quiet")
    assert "levels written" in out
    recs = records(tri)
    assert len(recs) == len(LADDER)                       # the exported ladder
    for (ratio, NT, NP), want in zip(recs, LADDER):
        assert want <= NT <= want + 2 and abs(ratio - 1200 / 1381) < 1e-6   # RATIO = image w / h (main.cpp:54)
    # every level is a consistent triangulation of the whole domain
    ctx = capi.Context(0, 800, 920)
    for level in (0, len(LADDER) // 2, len(LADDER) - 1):
        ratio, tris, he, cols, pts, org = read_level(tri, level)
        check_halfedges(tris, he)
        assert tris[:, :3].min() >= 0 and tris[:, :3].max() < pts.shape[0]
        assert np.array_equal(pts, org)                    # export sets originpoints = points (main.cpp:226)
    # the final state, one more grad-iter: HIP == oracle, bit for bit
    ratio, tris, he, cols, pts, org = read_level(tri, len(LADDER) - 1)
    img = synth.voronoi_raster(800, 920, seed=5, sites=60)
    ctx.set_ratio(ratio)
    ctx.set_image(capi.IMAGE_A, img)
    ctx.upload(pts, tris)
    ctx.iterate(capi.default_params(0), 1)
    ref = O.iterate(img, pts, tris, 0, ratio, 0.00005, 1, literal=False)
    assert np.array_equal(ctx.retrieve(capi.BUF_TENERGY), ref["ten"])
    assert np.array_equal(ctx.retrieve(capi.BUF_COLNUM), ref["cn"])
    assert np.array_equal(ctx.retrieve(capi.BUF_POINTS).view(np.uint32), ref["points"].view(np.uint32))
    base = ctx.retrieve(capi.BUF_COLNUM)[: tris.shape[0]]
    assert int(base.sum()) >= 800 * 920                    # the base triangles cover the raster (folded ones count twice)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["meninas", "synthetic"])
def test_config2_shortcuts_decide_like_the_literal_frame(tmp_path, kind):
    """config 2 to 1000 triangles twice on the HIP path: with the harness's shortcuts (entries the host looks at only,
    filtered sweeps, radix ranking) and with `-literal` (the frame as the reference writes it): identical .tri bytes"""
    ppm = str(tmp_path / "m.ppm")
    write_ppm(ppm, config2_picture(kind))
    gpu = build_gpu("triangulate")
    # (on the picture itself a split takes ~10 000 frames: its first 40 000 frames, levels at 4 and 6 triangles)
    args = ["-i", ppm, "-levels", "50,100,200,400,700,1000" if kind == "synthetic" else "4,6", "-window", "1.5", "-quiet"] + \
           ([] if kind == "synthetic" else ["-maxframes", "40000"])
    o1 = run(gpu, *args, "-o", str(tmp_path / "s.tri"))
    o2 = run(gpu, *args, "-literal", "-o", str(tmp_path / "l.tri"))
    assert o1.replace("s.tri", "X") == o2.replace("l.tri", "X") and ("levels written 6" in o1 or kind != "synthetic")
    assert len(records(str(tmp_path / "s.tri"))) >= 1
    assert open(str(tmp_path / "s.tri"), "rb").read() == open(str(tmp_path / "l.tri"), "rb").read()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["meninas", "synthetic"])
def test_config2_schedule_bytes_match_oracle_backend(tmp_path, kind):
    """the same schedule with a frame cap, HIP against the oracle-backed C ABI: identical .tri bytes (the topology
    decisions depend on every energy bit)"""
    ppm = str(tmp_path / "m.ppm")
    write_ppm(ppm, config2_picture(kind))
    cpu, gpu = build_cpu("triangulate"), build_gpu("triangulate")
    # (on the picture itself the first convergence comes after some hundred frames and a split takes thousands: its first level only)
    args = ["-i", ppm, "-levels", "6,12,20" if kind == "synthetic" else "2,4", "-window", "1.5", "-maxframes", "500" if kind == "synthetic" else "1200", "-quiet"]
    o1 = run(cpu, *args, "-o", str(tmp_path / "c.tri"))
    o2 = run(gpu, *args, "-o", str(tmp_path / "g.tri"))
    assert o1 == o2
    c, g = open(str(tmp_path / "c.tri"), "rb").read(), open(str(tmp_path / "g.tri"), "rb").read()
    assert c == g and len(records(str(tmp_path / "g.tri"))) >= 1


def two_views(kind):
    """config 3's two views at the window of resource/imageA.png / imageB.png (1200x675 / 1.5 = 800x450): the pictures themselves (through the
    integer resampler of tpose_amd/photos.py), config 5's shoeA / shoeB (960x540 / 1.5 = 640x360), or the synthetic pair of rounds 1-5"""
    if kind == "photo":
        return photos.window("imageA"), photos.window("imageB")
    if kind == "shoes":
        return photos.window("shoeA"), photos.window("shoeB")
    A = photo_like(800, 450, 77, 120)
    return A, synth.displaced_raster(A, amp=8.0)


@pytest.fixture(scope="module", params=["photo", "synthetic"])
def config3(tmp_path_factory, request):
    """two views at the window of resource/imageA.png / imageB.png (1200x675 / 1.5 = 800x450) with their 5-level
    hierarchies (50 ... 400 triangles) from the triangulate harness on the HIP path"""
    d = tmp_path_factory.mktemp("config3_" + request.param)
    A, B = two_views(request.param)
    assert A.shape == B.shape == ((450, 800, 4))
    write_ppm(str(d / "a.ppm"), A)
    write_ppm(str(d / "b.ppm"), B)
    gpu = build_gpu("triangulate")
    for n in ("a", "b"):
        run(gpu, "-i", str(d / (n + ".ppm")), "-o", str(d / (n + ".tri")), "-levels", "50,100,200,300,400", "-quiet")
        assert len(records(str(d / (n + ".tri")))) == 5
    return d


@pytest.mark.gpu
@pytest.mark.parametrize("schedule", ["as_written", "two_way", "mutual"])
def test_config3_five_level_warp_bytes_match_oracle_backend(config3, schedule):
    """config 3: 5-level coarse-to-fine warp between the two views, every schedule, HIP against the oracle backend"""
    import shutil
    cpu, gpu = build_cpu("warp"), build_gpu("warp")
    outs = []
    for tag, exe in (("c", cpu), ("g", gpu)):
        for n in ("a", "b"):
            shutil.copy(str(config3 / (n + ".tri")), str(config3 / ("%s_%s_%s.tri" % (tag, schedule, n))))
        ta, tb = (str(config3 / ("%s_%s_%s.tri" % (tag, schedule, n))) for n in ("a", "b"))
        outs.append(run(exe, "-ia", str(config3 / "a.ppm"), "-ib", str(config3 / "b.ppm"), "-ta", ta, "-tb", tb,
                        "-schedule", schedule, "-levelframes", "60", "-quiet"))
    assert outs[0] == outs[1] and "levels 5" in outs[0]
    for n in ("a", "b"):
        c = open(str(config3 / ("c_%s_%s.tri.warp" % (schedule, n))), "rb").read()
        g = open(str(config3 / ("g_%s_%s.tri.warp" % (schedule, n))), "rb").read()
        assert c == g and len(records(str(config3 / ("g_%s_%s.tri.warp" % (schedule, n))))) == 5


@pytest.mark.gpu
def test_config3_two_gpu_driver_matches_single_gpu(config3):
    """config 3 through the C++ two-rank driver (one image per rank): byte-identical to `warp -schedule mutual`"""
    import shutil
    gpu, gpu2 = build_gpu("warp"), build_gpu("warp2")
    for who in ("one", "two"):
        for n in ("a", "b"):
            shutil.copy(str(config3 / (n + ".tri")), str(config3 / ("m_%s_%s.tri" % (who, n))))
    common = ["-ia", str(config3 / "a.ppm"), "-ib", str(config3 / "b.ppm"), "-levelframes", "400", "-quiet"]
    ta, tb = (str(config3 / ("m_one_%s.tri" % n)) for n in ("a", "b"))
    run(gpu, *common, "-ta", ta, "-tb", tb, "-schedule", "mutual")
    ta2, tb2 = (str(config3 / ("m_two_%s.tri" % n)) for n in ("a", "b"))
    procs = [subprocess.Popen([gpu2, "-rank", str(r), "-idfile", str(config3 / "link"), "-transport", "fifo", "-device", "0", *common,
                               "-ta", ta2, "-tb", tb2], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in (0, 1)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    for one, two in ((ta, ta2), (tb, tb2)):
        assert open(one + ".warp", "rb").read() == open(two + ".warp", "rb").read()
        assert len(records(one + ".warp")) == 5


@pytest.fixture(scope="module")
def shoes(tmp_path_factory):
    """config 5's own pictures: resource/shoeA.png / shoeB.png at the reference's window (960x540 / 1.5 = 640x360) with their 5-level
    hierarchies from the triangulate harness on the HIP path"""
    d = tmp_path_factory.mktemp("config5_shoes")
    A, B = two_views("shoes")
    assert A.shape == B.shape == ((360, 640, 4))
    write_ppm(str(d / "a.ppm"), A)
    write_ppm(str(d / "b.ppm"), B)
    gpu = build_gpu("triangulate")
    for n in ("a", "b"):
        run(gpu, "-i", str(d / (n + ".ppm")), "-o", str(d / (n + ".tri")), "-levels", "50,100,200,300,400", "-quiet")
        assert len(records(str(d / (n + ".tri")))) == 5
    return d


@pytest.mark.gpu
def test_config5_on_shoeA_shoeB(shoes):
    """config 5 on the pictures it names: hierarchies of shoeA / shoeB -> two-way warp on the HIP path -> correspondences -> F"""
    _config5(shoes)


@pytest.mark.gpu
def test_config5_fundamental_matrix_from_config3_warp(config3):
    """config 5 (host side, like the reference): correspondences from the warped vertices of config 3's finest level ->
    F_Sampson / F_LMEDS / F_RANSAC; the synthetic views differ by a smooth displacement, so the epipolar fit is loose but
    must be finite, and the Sampson-refined F must not be worse than the plain RANSAC estimate it starts from"""
    _config5(config3)


def _config5(config3):
    import shutil
    gpu = build_gpu("warp")
    for n in ("a", "b"):
        shutil.copy(str(config3 / (n + ".tri")), str(config3 / ("f_%s.tri" % n)))
    ta, tb = str(config3 / "f_a.tri"), str(config3 / "f_b.tri")
    run(gpu, "-ia", str(config3 / "a.ppm"), "-ib", str(config3 / "b.ppm"), "-ta", ta, "-tb", tb, "-schedule", "two_way",
        "-levelframes", "200", "-quiet")
    subprocess.check_call(["make", "-s", "-C", HOST, "fundamental"])
    dump = str(config3 / "matches.txt")
    out = run(os.path.join(HOST, "fundamental"), ta, ta + ".warp", tb, tb + ".warp", "-dumpmatches", dump)
    vals = [float(l.split(":")[1]) for l in out.splitlines() if "mean squared Sampson distance" in l]
    assert len(vals) == 3 and all(np.isfinite(v) for v in vals)
    ma = int(out.split("Found A Matches: ")[1].split()[0]); mb = int(out.split("Found B Matches: ")[1].split()[0])
    assert ma > 20 and mb > 20
    # "Sampson error vs reference" (BASELINE config 5): the NumPy restatement of source/multiview.hpp:187-242 (tests/
    # test_multiview.py) on the very matches the harness used.  Tolerances as in the CPU test of the same comparison: 1 % of
    # the mean squared Sampson distance (float32 storage between the 100 re-weighting rounds), entries up to scale 5e-3.
    from test_multiview import np_fsampson, np_mean_sampson, same_up_to_scale
    M = np.loadtxt(dump, dtype=np.float64)
    A, B = M[:, 0:2].astype(np.float32), M[:, 2:4].astype(np.float32)
    assert A.shape[0] == ma + mb
    FS = np.loadtxt(dump + ".F", dtype=np.float64)
    Fn = np_fsampson(A, B)
    s_mine, s_np = np_mean_sampson(FS, A, B), np_mean_sampson(Fn, A, B)
    assert abs(s_mine - vals[0]) <= 1e-4 * s_mine + 1e-15        # what the harness printed is this F on these matches
    assert s_mine <= s_np * 1.01 + 1e-14
    assert same_up_to_scale(FS.astype(np.float32), Fn, 5e-3)
