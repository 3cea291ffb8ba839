#!/usr/bin/env python
"""BASELINE config 3 on synthetic data: 5-level coarse-to-fine two-way warp between two views at the window size the
reference would use for resource/imageA.png / imageB.png (1200x675 / 1.5 = 800x450).
  (a) the single-GPU harness (tpose_amd/host/warp, both schedules), reference frame loop with readbacks;
  (b) tpose_amd.warp_dist: one direction per rank, fused iterations, per-level exchange -- two ranks, here both on
      GPU 0 over gloo (a 1-GPU box; on a node it is `--nproc-per-node 2` over RCCL).
Needs an MI355X."""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from tpose_amd import synth  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tpose_amd", "host")
out = os.path.join(ROOT, "gpurun_out")
os.makedirs(out, exist_ok=True)
W, H = 800, 450
A = synth.voronoi_raster(W, H, seed=77, sites=120)
rgb = A[:, :, :3].astype(np.float32)
A[:, :, :3] = np.clip(128.0 + (rgb - 128.0) * 0.1 + 0.5, 0, 255).astype(np.uint8)  # photograph-like contrast (run_config2.py)
B = synth.displaced_raster(A, amp=8.0)


def ppm(path, img):
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(np.ascontiguousarray(img[:, :, :3]).tobytes())


pa, pb = os.path.join(out, "c3a.ppm"), os.path.join(out, "c3b.ppm")
ppm(pa, A); ppm(pb, B)
subprocess.check_call(["make", "-s", "-C", HOST, "triangulate", "warp", "libtpose_host.so"])
levels = "50,100,200,300,400"
for n, p in (("a", pa), ("b", pb)):
    t = os.path.join(out, "c3%s.tri" % n)
    if os.path.exists(t):
        os.remove(t)
    r = subprocess.run([os.path.join(HOST, "triangulate"), "-i", p, "-o", t, "-levels", levels, "-quiet"],
                       capture_output=True, text=True, timeout=300)
    print("hierarchy", n, r.stdout.strip().splitlines()[-1], "|", r.stderr.strip().splitlines()[-1])
for schedule in ("as_written", "two_way"):
    for n in ("a", "b"):
        w = os.path.join(out, "c3%s.tri.warp" % n)
        if os.path.exists(w):
            os.remove(w)
    t0 = time.perf_counter()
    r = subprocess.run([os.path.join(HOST, "warp"), "-ia", pa, "-ib", pb, "-ta", os.path.join(out, "c3a.tri"), "-tb",
                        os.path.join(out, "c3b.tri"), "-schedule", schedule, "-levelframes", "4000", "-quiet"],
                       capture_output=True, text=True, timeout=600)
    print("harness", schedule, (r.stdout.strip().splitlines() or ["?"])[-1], "| %.2f s" % (time.perf_counter() - t0))
env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
for n in ("a", "b"):
    w = os.path.join(out, "c3%s.tri.warp" % n)
    if os.path.exists(w):
        os.remove(w)
t0 = time.perf_counter()
r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", "29671", "-m", "tpose_amd.warp_dist", "--ia", pa, "--ib", pb, "--ta", os.path.join(out, "c3a.tri"),
                    "--tb", os.path.join(out, "c3b.tri"), "--frames", "4000", "--backend", "gloo", "--share-gpu"],
                   capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
print("warp_dist 2 ranks (shared GPU, gloo):", [l for l in r.stdout.splitlines() if "level" in l][-3:], "| %.2f s incl. start-up" % (time.perf_counter() - t0))
if r.returncode:
    print(r.stderr[-2000:])
