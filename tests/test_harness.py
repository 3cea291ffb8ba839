"""Schedule-level checks of the headless harnesses (tpose_amd/host): the reference's frame schedule
(software/triangulate/main.cpp:190-353, software/warp/main.cpp:214-283) through the C++ host mirror.

The same harness source is built twice -- against libtpose_hip.so (GPU) and against a test-only
CPU backend that implements the C ABI on the oracle -- and the .tri files must agree byte for byte."""
import os
import subprocess

import numpy as np
import pytest

from tpose_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HOST = os.path.join(ROOT, "tpose_amd", "host")
BUILD = os.path.join(HERE, "_build")


def write_ppm(path, img):
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(np.ascontiguousarray(img[:, :, :3]).tobytes())


def build_cpu(name, flags=()):
    """harness `name` linked against the oracle-backed C ABI (tests only)"""
    os.makedirs(BUILD, exist_ok=True)
    obj = os.path.join(BUILD, "oracle_backend.o")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-c", os.path.join(HERE, "host", "oracle_backend.c"), "-o", obj])
    ora = os.path.join(BUILD, "tp_oracle.o")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-c", os.path.join(ROOT, "oracle", "tp_oracle.c"), "-o", ora])
    exe = os.path.join(BUILD, name + "_cpu")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-I" + HOST] + list(flags) +
                          [os.path.join(HOST, name + ".cpp"), obj, ora, "-fopenmp", "-lm", "-pthread", "-o", exe])
    return exe


def build_gpu(name):
    subprocess.check_call(["make", "-s", "-C", HOST, name])
    return os.path.join(HOST, name)


def run(exe, *args):
    out = subprocess.run([exe] + list(args), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    return out.stdout


def records(path):
    """parse a stacked .tri file -> list of (ratio, NT, NP)"""
    data = open(path, "rb").read()
    recs, off = [], 0
    while off < len(data):
        ratio = np.frombuffer(data, np.float32, 1, off)[0]
        NT = int(np.frombuffer(data, np.int32, 1, off + 4)[0])
        NP = int(np.frombuffer(data, np.int32, 1, off + 8 + 36 * NT)[0])
        off += 8 + 36 * NT + 4 + 16 * NP
        recs.append((float(ratio), NT, NP))
    assert off == len(data)
    return recs


@pytest.fixture(scope="module")
def scene(tmp_path_factory):
    d = tmp_path_factory.mktemp("harness")
    img = synth.voronoi_raster(160, 120, seed=21, sites=7, noise=3)
    imgB = synth.displaced_raster(img, amp=4.0)
    write_ppm(str(d / "a.ppm"), img)
    write_ppm(str(d / "b.ppm"), imgB)
    return d


def test_triangulate_schedule_on_cpu_backend(scene):
    """host logic only (no GPU): the schedule runs, splits, exports the 50-triangle level"""
    exe = build_cpu("triangulate")
    out = run(exe, "-i", str(scene / "a.ppm"), "-o", str(scene / "cpu.tri"), "-maxframes", "1500", "-maxtris", "60", "-quiet")
    assert "levels written" in out
    recs = records(str(scene / "cpu.tri"))
    assert len(recs) >= 1 and recs[0][1] >= 50 and abs(recs[0][0] - 160 / 120) < 1e-6


def test_triangulate_shortcuts_decide_like_the_literal_frame(scene):
    """the harness reads back only the entries the host looks at, filters the per-frame angle / collapse sweeps with products
    and ranks the flip set with a radix sort; `-literal` runs the frame as the reference writes it (13 NT entries, every
    arc cosine, comparison sort): same stdout (frames, triangles, energies per change), same .tri bytes"""
    exe = build_cpu("triangulate")
    args = ["-i", str(scene / "a.ppm"), "-maxframes", "2500", "-maxtris", "90", "-levels", "10,20,30,50,70,90"]
    o1 = run(exe, *args, "-o", str(scene / "short.tri"))
    o2 = run(exe, *args, "-literal", "-o", str(scene / "lit.tri"))
    assert o1.replace("short.tri", "X") == o2.replace("lit.tri", "X") and "levels written" in o1
    assert open(str(scene / "short.tri"), "rb").read() == open(str(scene / "lit.tri"), "rb").read()
    assert len(records(str(scene / "short.tri"))) >= 3


@pytest.mark.gpu
def test_triangulate_gpu_matches_cpu_backend_bytes(scene):
    cpu, gpu = build_cpu("triangulate"), build_gpu("triangulate")
    args = ["-i", str(scene / "a.ppm"), "-maxframes", "1500", "-maxtris", "60", "-quiet"]
    o1 = run(cpu, *args, "-o", str(scene / "c.tri"))
    o2 = run(gpu, *args, "-o", str(scene / "g.tri"))
    assert o1 == o2
    assert open(str(scene / "c.tri"), "rb").read() == open(str(scene / "g.tri"), "rb").read()


@pytest.mark.gpu
@pytest.mark.parametrize("schedule", ["as_written", "two_way"])
def test_warp_gpu_matches_cpu_backend_bytes(scene, schedule):
    # hierarchy for both views from the triangulate harness (CPU backend), then warp both ways
    cpu_t = build_cpu("triangulate")
    for n in ("a", "b"):
        run(cpu_t, "-i", str(scene / (n + ".ppm")), "-o", str(scene / (n + ".tri")), "-maxframes", "2500", "-maxtris", "110", "-quiet")
        assert len(records(str(scene / (n + ".tri")))) >= 2
    cpu, gpu = build_cpu("warp"), build_gpu("warp")
    outs = []
    for tag, exe in (("c", cpu), ("g", gpu)):
        for n in ("a", "b"):
            data = open(str(scene / (n + ".tri")), "rb").read()
            open(str(scene / ("%s_%s_%s.tri" % (tag, schedule, n))), "wb").write(data)
        ta, tb = (str(scene / ("%s_%s_%s.tri" % (tag, schedule, n))) for n in ("a", "b"))
        outs.append(run(exe, "-ia", str(scene / "a.ppm"), "-ib", str(scene / "b.ppm"), "-ta", ta, "-tb", tb,
                        "-schedule", schedule, "-levelframes", "150", "-quiet"))
        assert os.path.exists(ta + ".warp") and os.path.exists(tb + ".warp")
    assert outs[0] == outs[1]
    for n in ("a", "b"):
        c = open(str(scene / ("c_%s_%s.tri.warp" % (schedule, n))), "rb").read()
        g = open(str(scene / ("g_%s_%s.tri.warp" % (schedule, n))), "rb").read()
        assert c == g and len(c) > 0


@pytest.mark.gpu
def test_view_gpu_matches_cpu_backend_bytes(scene):
    """headless software/view: the picture of a .tri level (stored colours, morphed by s) is the same
    PPM from the HIP renderer and from the oracle's per-pixel coverage test"""
    cpu_t = build_cpu("triangulate")
    run(cpu_t, "-i", str(scene / "a.ppm"), "-o", str(scene / "v.tri"), "-maxframes", "1500", "-maxtris", "60", "-quiet")
    cpu, gpu = build_cpu("view"), build_gpu("view")
    for s in ("0", "0.37"):
        o1 = run(cpu, "-t", str(scene / "v.tri"), "-s", s, "-height", "240", "-o", str(scene / "vc.ppm"))
        o2 = run(gpu, "-t", str(scene / "v.tri"), "-s", s, "-height", "240", "-o", str(scene / "vg.ppm"))
        assert o1.replace("vc.ppm", "") == o2.replace("vg.ppm", "")
        a, b = open(str(scene / "vc.ppm"), "rb").read(), open(str(scene / "vg.ppm"), "rb").read()
        assert len(a) > 240 * 240 * 3 and a == b


def test_fundamental_harness_on_cpu(scene):
    """BASELINE config 5 plumbing, host only: hierarchy for two views -> warp -> correspondences from the
    warped vertices -> F_Sampson / F_RANSAC; and the reference's own match file"""
    cpu_t, cpu_w = build_cpu("triangulate"), build_cpu("warp")
    for n in ("a", "b"):
        run(cpu_t, "-i", str(scene / (n + ".ppm")), "-o", str(scene / ("f_%s.tri" % n)), "-maxframes", "1500", "-maxtris", "60", "-quiet")
    ta, tb = str(scene / "f_a.tri"), str(scene / "f_b.tri")
    run(cpu_w, "-ia", str(scene / "a.ppm"), "-ib", str(scene / "b.ppm"), "-ta", ta, "-tb", tb, "-levelframes", "100", "-quiet")
    subprocess.check_call(["make", "-s", "-C", HOST, "fundamental"])
    exe = os.path.join(HOST, "fundamental")
    out = run(exe, ta, ta + ".warp", tb, tb + ".warp", "-points", str(scene / "pts.txt"))
    na, nb = records(ta)[0][2], records(tb)[0][2]
    ma = int(out.split("Found A Matches: ")[1].split()[0]); mb = int(out.split("Found B Matches: ")[1].split()[0])
    assert na - 4 <= ma <= na and nb - 4 <= mb <= nb  # every vertex some triangle uses
    vals = [float(l.split(":")[1]) for l in out.splitlines() if "mean squared Sampson distance" in l]
    assert len(vals) == 3 and all(np.isfinite(v) for v in vals) and vals[0] < 1e-2
    pts = np.loadtxt(str(scene / "pts.txt"))
    assert pts.shape == (ma + mb, 3)
    out = run(exe, "-matches", os.path.join(HERE, "golden", "sfm_matches.txt"), "-image", "960x540")
    vals = [float(l.split(":")[1]) for l in out.splitlines() if "mean squared Sampson distance" in l]
    assert len(vals) == 3 and all(np.isfinite(v) and v < 1e-3 for v in vals)


# ---- the two-GPU C++ driver (tpose_amd/host/warp2.cpp) --------------------------------------------------------------
def _hierarchies(scene, build):
    for n in ("a", "b"):
        if not os.path.exists(str(scene / ("h_%s.tri" % n))):
            run(build, "-i", str(scene / (n + ".ppm")), "-o", str(scene / ("h_%s.tri" % n)), "-maxframes", "2500", "-maxtris", "110", "-quiet")
        assert len(records(str(scene / ("h_%s.tri" % n)))) >= 2


def _mutual_vs_two_ranks(scene, warp_exe, warp2_exe, tag, extra=(), transport="fifo", per_rank=lambda r: []):
    """`warp -schedule mutual` (one process, four descents per level) against two `warp2` processes exchanging their
    meshes: the .tri.warp files must be the same bytes"""
    import shutil
    for who in ("one", "two"):
        for n in ("a", "b"):
            shutil.copy(str(scene / ("h_%s.tri" % n)), str(scene / ("%s_%s_%s.tri" % (tag, who, n))))
    common = ["-ia", str(scene / "a.ppm"), "-ib", str(scene / "b.ppm"), "-levelframes", "120", "-quiet"]
    ta, tb = (str(scene / ("%s_one_%s.tri" % (tag, n))) for n in ("a", "b"))
    run(warp_exe, *common, "-ta", ta, "-tb", tb, "-schedule", "mutual")
    ta2, tb2 = (str(scene / ("%s_two_%s.tri" % (tag, n))) for n in ("a", "b"))
    idfile = str(scene / ("%s_link" % tag))
    procs = [subprocess.Popen([warp2_exe, "-rank", str(r), "-idfile", idfile, "-transport", transport, *common, "-ta", ta2, "-tb", tb2, *extra, *per_rank(r)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in (0, 1)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "rank 0 frames" in outs[0] and "rank 1 frames" in outs[1]
    for one, two in ((ta, ta2), (tb, tb2)):
        a, b = open(one + ".warp", "rb").read(), open(two + ".warp", "rb").read()
        assert len(a) > 0 and a == b
    assert len(records(ta + ".warp")) == len(records(str(scene / "h_a.tri"))) or len(records(ta + ".warp")) == len(records(str(scene / "h_b.tri")))


def _mutual_vs_banded(scene, warp_exe, warp2_exe, tag, frames, extra=(), fixed=True):
    """`warp -schedule mutual -fixedframes` (one process) against FOUR `warp2 -bands 2` processes -- two bands for each of the two
    directions: both bands of a direction must write the unsplit run's bytes"""
    import shutil
    for who in ("one", "four"):
        for n in ("a", "b"):
            shutil.copy(str(scene / ("h_%s.tri" % n)), str(scene / ("%s_%s_%s.tri" % (tag, who, n))))
    common = ["-ia", str(scene / "a.ppm"), "-ib", str(scene / "b.ppm"), "-levelframes", str(frames), "-quiet"] + (["-fixedframes"] if fixed else [])
    ta, tb = (str(scene / ("%s_one_%s.tri" % (tag, n))) for n in ("a", "b"))
    run(warp_exe, *common, "-ta", ta, "-tb", tb, "-schedule", "mutual")
    ta4, tb4 = (str(scene / ("%s_four_%s.tri" % (tag, n))) for n in ("a", "b"))
    idfile = str(scene / ("%s_blink" % tag))
    procs = [subprocess.Popen([warp2_exe, "-rank", str(r), "-bands", "2", "-band", str(b), "-idfile", idfile, "-transport", "fifo", *common,
                               "-ta", ta4, "-tb", tb4, *extra], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in (0, 1) for b in (0, 1)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    for one, four in ((ta, ta4), (tb, tb4)):
        a = open(one + ".warp", "rb").read()
        assert len(a) > 0 and a == open(four + ".warp", "rb").read() and a == open(four + ".warp.band1", "rb").read()
    return outs


def test_warp2_bands_plumbing_on_cpu_backend(scene):
    """world size 4 without a GPU: two bands per direction over named pipes (mate links, per-band cross links, stop-together);
    the oracle-backed C ABI runs every band's descents whole, so this pins the harness, not the seam"""
    _hierarchies(scene, build_cpu("triangulate"))
    _mutual_vs_banded(scene, build_cpu("warp"), build_cpu("warp2", flags=["-DWARP2_NO_RCCL"]), "cpub", 40)


@pytest.mark.gpu
def test_warp2_bands_split_every_descent_over_two_processes(scene):
    """SURVEY section 8 row e3 on the HIP path: every direction's descents run as two bands in two processes that map each
    other's mailbox (hipIpcGetMemHandle) and hand vertex positions over on the device, grad-iter by grad-iter; all four share
    GPU 0 here (a proxy for four GPUs: the same protocol without the links), 4 patches per band.  Bytes equal the unsplit
    run's, and the launches were persistent and none gave up."""
    import re
    _hierarchies(scene, build_cpu("triangulate"))
    outs = _mutual_vs_banded(scene, build_gpu("warp"), build_gpu("warp2"), "gpub", 150, extra=["-device", "0", "-bandpatches", "4"])
    # ... and with the schedule's own convergence test (tp_iterate_until on bands: every band's host tests all bands' energies)
    outs += _mutual_vs_banded(scene, build_gpu("warp"), build_gpu("warp2"), "gpuc", 150, extra=["-device", "0", "-bandpatches", "4"], fixed=False)
    for o in outs:
        m = re.search(r"persistent launches (\d+), patches of the plan (\d+), launches given up (\d+)", o)
        assert m, o
        assert int(m.group(1)) > 0 and int(m.group(2)) == 8 and int(m.group(3)) == 0, o


def test_warp2_two_ranks_match_single_process_on_cpu_backend(scene):
    """world size 2 without a GPU: the C++ two-rank driver over named pipes, against the oracle-backed C ABI"""
    cpu_t = build_cpu("triangulate")
    _hierarchies(scene, cpu_t)
    _mutual_vs_two_ranks(scene, build_cpu("warp"), build_cpu("warp2", flags=["-DWARP2_NO_RCCL"]), "cpu")


@pytest.mark.gpu
def test_warp2_over_rccl_on_two_gpus(scene):
    """the hand-over and stop-together protocol over ncclSend / ncclRecv between two GPUs (skipped on a one-GPU box: RCCL
    refuses two ranks on one device); a stale id file of an earlier run lies in the way and is told apart by the nonce"""
    from tpose_amd import capi
    if capi.device_count() < 2:
        pytest.skip("needs two GPUs")
    _hierarchies(scene, build_cpu("triangulate"))
    open(str(scene / "rccl_link"), "wb").write(b"\0" * 136)   # (nonce 0 + a dead unique id)
    _mutual_vs_two_ranks(scene, build_gpu("warp"), build_gpu("warp2"), "rccl", transport="rccl", extra=["-nonce", str(os.getpid())],
                         per_rank=lambda r: ["-device", str(r)])


@pytest.mark.gpu
def test_warp2_two_ranks_match_single_process_on_gpu(scene):
    """the same on the HIP path: two ranks share GPU 0 (RCCL refuses two ranks on one device, so the pipes carry the
    meshes here), and the bytes also equal the CPU backend's"""
    cpu_t = build_cpu("triangulate")
    _hierarchies(scene, cpu_t)
    _mutual_vs_two_ranks(scene, build_gpu("warp"), build_gpu("warp2"), "gpu", extra=["-device", "0"])
    _mutual_vs_two_ranks(scene, build_cpu("warp"), build_cpu("warp2", flags=["-DWARP2_NO_RCCL"]), "cpu")
    for n in ("a", "b"):
        assert open(str(scene / ("gpu_one_%s.tri.warp" % n)), "rb").read() == open(str(scene / ("cpu_one_%s.tri.warp" % n)), "rb").read()


@pytest.mark.gpu
def test_warp2_rccl_self_exchange(scene):
    """RCCL itself from the C++ driver: communicator, stream, device buffers, one grouped ncclSend / ncclRecv"""
    exe = build_gpu("warp2")
    out = run(exe, "-selftest", "-idfile", str(scene / "rccl_id"))
    assert "RCCL self exchange OK" in out
