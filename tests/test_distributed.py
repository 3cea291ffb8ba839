"""world_size-2 gloo test (CPU) of the multi-GPU path: the two-way warp driver's per-level exchange of
vertex buffers, pairing, and the bench's max-over-ranks reduction."""
import json
import os
import socket
import subprocess
import sys

import numpy as np

from tpose_amd import hostlib, synth, warp_dist

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def write_stack(path, img, ratio, grids):
    """a stacked .tri hierarchy (coarse to fine) with stored colours, originpoints = points"""
    if os.path.exists(path):
        os.remove(path)
    hostlib.set_ratio(ratio)
    t = hostlib.Triangulation()
    for g in grids:
        pts, tris, he = synth.grid_triangulation(g[0], g[1], ratio=ratio, jitter=0.15)
        t.assign(tris, pts, pts, he, synth.mean_colors(img, pts, tris, ratio))
        t.write(path)


def test_pack_unpack_roundtrip():
    hostlib.set_ratio(1.5)
    t = hostlib.Triangulation()
    t.split(0)
    u = warp_dist.unpack(warp_dist.pack(t))
    assert u.NT == t.NT and u.NP == t.NP
    assert np.array_equal(u.triangles, t.triangles) and np.array_equal(u.points, t.points)
    assert np.array_equal(u.originpoints, t.originpoints)


def test_two_way_warp_world_size_2_gloo(tmp_path):
    W, H = 96, 64
    ratio = float(np.float32(W) / np.float32(H))
    A = synth.voronoi_raster(W, H, seed=5, sites=6, noise=2)
    B = synth.displaced_raster(A, amp=3.0)
    np.save(str(tmp_path / "A.npy"), A)
    np.save(str(tmp_path / "B.npy"), B)
    write_stack(str(tmp_path / "A.tri"), A, ratio, [(3, 2), (6, 4)])
    write_stack(str(tmp_path / "B.tri"), B, ratio, [(3, 2), (6, 4)])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), str(tmp_path), "24"],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    outs = [json.load(open(str(tmp_path / ("out%d.json" % r)))) for r in range(2)]
    assert outs[0]["worst"] == outs[1]["worst"] == 11.0          # MAX over ranks
    assert outs[0]["seed"] != outs[1]["seed"]                    # one replica per rank
    for r, name in enumerate(("A.tri", "B.tri")):
        lv = outs[r]["levels"]
        assert [l["NT"] for l in lv] == [12, 48]                 # both hierarchy levels processed
        assert all(np.isfinite(l["residual"]) for l in lv)
        # one record per level appended to <tri>.warp, same sizes as the input stack
        data = open(str(tmp_path / (name + ".warp")), "rb").read()
        assert len(data) == os.path.getsize(str(tmp_path / name))
    # the finer level was read warped-on-read: its origin points are untouched, its points moved
    hostlib.set_ratio(ratio)
    t = hostlib.Triangulation()
    assert t.read(str(tmp_path / "A.tri.warp")) and t.read(str(tmp_path / "A.tri.warp"))
    assert t.NT == 48 and not np.array_equal(t.points, t.originpoints)


def test_band_split_world_size_2_gloo(tmp_path):
    """SURVEY section 8 row e3 on the CPU: two ranks, each replaying HALF of the patches of one descent (the persistent kernel's
    lane functions, tests/emul), positions crossing the seam through a gloo all_gather of the mailbox after every grad-iter.
    A rank that did not receive a position it needs fails with a tag mismatch; both end with the oracle's bits."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "band_worker.py"), str(tmp_path), "70"],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    for r in range(2):
        out = json.load(open(str(tmp_path / ("band%d.json" % r))))
        assert out["rc"] == 0 and out["patches"] == 8 and out["exchanges"] == 70 and out["same_as_oracle"], out


import pytest  # noqa: E402

ROOT = os.path.dirname(HERE)


def _torchrun(args, nproc, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port())] + args
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


@pytest.mark.gpu
def test_bench_line_contract_one_gpu():
    """`python bench.py` with the driver's flags: ONE JSON line with the metric of BASELINE.json, the roofline object of
    the dominant kernel (k_persist: K grad-iters per launch) and the CPU baseline; the kernel trace and the counter passes are collected live"""
    import sys
    r = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--warmup", "5"], capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["steps"] == 20 and line["warmup"] == 5 and line["higher_is_better"] is True
    assert line["unit"] == "triangles*grad-iters/s" and line["vs_baseline"] is None
    # round 6: the headline workload is the reference's own picture (meninas.png at the metric's 2048 x 2048), the synthetic raster's figures beside it
    assert line["data"].startswith("reference photograph") and "meninas" in line["data"] and "meninas" in line["config"]["workload"]
    ph = line["ms_per_step_on_reference_photos"]
    assert abs(ph["meninas"] - line["ms_per_step"]) < 1e-12 and all(isinstance(ph[k], float) and ph[k] > 0 for k in ("fruit", "imageA", "shoeA"))
    assert line["value_synthetic_contrast_0.10"] > 0 and 0 < line["roofline_frac_synthetic_contrast_0.10"] < 1
    assert set(line["ms_per_step_by_contrast"]) == {"0.10", "0.14", "0.30", "1.00"}
    assert line["value_all_13_variants"] > 0 and 0 < line["roofline_frac_all_13_variants_by_bench_clock"] < 1
    assert abs(line["value"] - 3000 * 20 / (line["ms_per_step"] * 20e-3)) < 1e-6 * line["value"]
    assert line["config"]["raster"] == [2048, 2048] and line["config"]["triangles"] == 3000 and "workload" in line["config"]
    rf = line["roofline"]
    assert rf["bound"] == "latency" and rf["nominal_bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and rf["kernel"] == "k_persist"
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0 < rf["frac"] < 1
    assert abs(rf["achieved"] - rf["algorithmic_bytes"] / (rf["kernel_us"] * 1e-6) / 1e9) < 1e-6 * rf["achieved"]
    assert rf["algorithmic_bytes"] == rf["algorithmic_bytes_per_grad_iter"] * rf["grad_iters_per_launch"]
    assert rf["traffic"] is None or rf["traffic"] > 0   # (far below the algorithmic bytes: the records a lane needs stay in registers)
    assert rf["observed_bound"] and line["ms_per_step_two_kernel_path"] > 0 and "persistent launches" in line["config"]["path"]
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["single_thread_value"] > 0 and cb["cpu_model"]
    assert line["ms_per_step_readback_every_iter"] > line["ms_per_step"]
    # what the headline does NOT show: SURVEY 8(d)'s raster as written, and all 13 variants formed in every grad-iter
    assert isinstance(line["ms_per_step_full_contrast"], float) and line["ms_per_step_full_contrast"] > 0
    assert isinstance(line["ms_per_step_all_13_variants"], float) and line["ms_per_step_all_13_variants"] >= line["ms_per_step"] * 0.9
    assert "12 displaced variants" in line["config"]["path"]
    cc = line["cold_cache"]   # SURVEY 8d caveat (iii): more tables than the Infinity Cache holds, cycled
    assert "error" not in cc, cc
    assert cc["contexts"] >= 5 and cc["raster_and_table_bytes_cycled"] > 256 * 2 ** 20
    assert cc["ms_per_step"] > 0 and cc["ms_per_step_same_loop_one_context"] > 0


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu():
    """the driver's launch line for N > 1 (one rank per GPU over RCCL), here with both ranks on the one GPU of the
    test box and gloo for the two collectives: one JSON line from rank 0, whole-job aggregate, weak scaling"""
    r = _torchrun(["bench.py", "--gpus", "2", "--steps", "256", "--warmup", "32", "--backend", "gloo", "--share-gpu"], 2)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 256 and line["warmup"] == 32
    assert line["value"] > 0 and abs(line["value"] - 3000 * 256 * 2 / (line["ms_per_step"] * 256e-3)) < 1e-6 * line["value"]
    assert "cpu_baseline" not in line and line["roofline"]["nominal_bound"] == "hbm" and 0 < line["roofline"]["frac"] < 1
    # beside the replicas: ONE pair on both ranks, here one direction cut into two bands, mailboxes shared between the
    # two processes through the ABI's export / import; no launch may have given up
    ps = line["pair_split"]
    assert "error" not in ps, ps
    assert ps["directions"] == 1 and ps["bands_per_direction"] == 2 and ps["launches_given_up_all_ranks"] == 0, ps
    assert ps["rank0"]["persist_iters"] >= 256 and ps["value"] > 0 and ps["speedup_vs_one_gpu"] > 0, ps
    # what makes a SCALE line checkable by itself: every rank took part in a SUM all-reduce, no band gave a launch up, every band's
    # mailbox is fine-grained memory -- at the top level of the line
    assert line["ranks_in_collective"] == 2 and line["collective_backend"] == "gloo"
    assert line["pair_split_launches_given_up_all_ranks"] == 0 and line["pair_split_mailbox_fine_grained"] is True and ps["mailbox_fine_grained"] is True


@pytest.mark.gpu
def test_two_way_warp_two_ranks_hip_engine(tmp_path):
    """tpose_amd.warp_dist end to end with the HIP engine: one direction per rank, per-level exchange"""
    W, H = 160, 120
    ratio = float(np.float32(W) / np.float32(H))
    A = synth.voronoi_raster(W, H, seed=5, sites=6, noise=2)
    B = synth.displaced_raster(A, amp=3.0)
    for name, img in (("A", A), ("B", B)):
        with open(str(tmp_path / (name + ".ppm")), "wb") as f:
            f.write(b"P6\n%d %d\n255\n" % (W, H))
            f.write(np.ascontiguousarray(img[:, :, :3]).tobytes())
        write_stack(str(tmp_path / (name + ".tri")), img, ratio, [(3, 2), (6, 4)])
    r = _torchrun(["-m", "tpose_amd.warp_dist", "--ia", str(tmp_path / "A.ppm"), "--ib", str(tmp_path / "B.ppm"), "--ta", str(tmp_path / "A.tri"),
                   "--tb", str(tmp_path / "B.tri"), "--frames", "64", "--backend", "gloo", "--share-gpu"], 2)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "levels: 2" in r.stdout
    for name in ("A.tri", "B.tri"):
        assert os.path.getsize(str(tmp_path / (name + ".warp"))) == os.path.getsize(str(tmp_path / name))


@pytest.mark.gpu
def test_batch_runner_two_ranks(tmp_path):
    """tools/run_batch.py (BASELINE config 4 in the small): pairs sharded over ranks, two directions of a pair
    concurrently in two contexts, per-pair metrics gathered at the end"""
    r = _torchrun([os.path.join("tools", "run_batch.py"), "--pairs", "3", "--iters", "48", "--size", "512", "--triangles", "150",
                   "--backend", "gloo", "--share-gpu"], 2)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and [p["pair"] for p in line["pairs"]] == [0, 1, 2]
    assert sorted(p["rank"] for p in line["pairs"]) == [0, 0, 1]
    for p in line["pairs"]:
        assert all(a < b for a, b in zip(p["energy_after"], p["energy_before"]))  # the descent lowers the warp energy
    assert line["triangles_iters_per_s"] > 0
