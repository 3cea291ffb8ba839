#!/usr/bin/env python
"""Replays under REAL contention, every kind of call: the random interleaving of tests/test_stress.py (short and long tp_iterate calls,
single frames, read-backs right behind calls, uploads of the moved mesh, tp_iterate_until, image and dp changes, the piecewise API) at
the metric size (2048^2 / 3000 triangles: full grids), once alone -- a digest of everything read back at every step -- and then again
while another process keeps the GPU full of its own persistent launches, so that launches of both are handed out half a grid each, give up
at random places of the sequence and are run again on the two-kernel path.  The digests must be the same, step by step.  (The oracle
checks the sequence itself in tests/test_stress.py; at this size it would take minutes per run, so here the uncontended run is the checker.)
  python tools/contention_soak.py [seeds]          needs an MI355X"""
import hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from tpose_amd import capi, synth

W = H = 2048

def hog():
    img, pts, tris, he, ratio = synth.workload(W, H, 3000, seed=99, contrast=0.1)
    p = capi.default_params(0)
    t_end = time.time() + float(sys.argv[2])
    launches = gave_up = 0
    while time.time() < t_end:
        # (a context that had a launch give up keeps to the two-kernel path from then on: a new one, so that the GPU stays full of grids)
        c = capi.Context(0, W, H); c.set_image(capi.IMAGE_A, img); c.upload(pts, tris, None); c.prepare(p)
        while time.time() < t_end and c.info(capi.INFO_PERSIST_FAILURES) == 0 and c.info(capi.INFO_CENSUS) == 1:
            for _ in range(8):
                c.iterate(p, 48)          # (no wait in between: the next launch is pending when this one ends)
            c.synchronize()
        launches += c.info(capi.INFO_PERSIST_LAUNCHES); gave_up += c.info(capi.INFO_PERSIST_FAILURES)
        c.close()
    print("hog: %d launches, %d given up" % (launches, gave_up), flush=True)

def sequence(seed, flavour):
    rng = np.random.default_rng(seed)
    img = synth.photo_contrast(synth.voronoi_raster(W, H, seed=40 + seed, sites=60), 0.1)
    img2 = synth.photo_contrast(synth.voronoi_raster(W, H, seed=90 + seed, sites=40), 0.1)
    ratio = 1.0
    pts, tris, _ = synth.grid_triangulation(50, 30, ratio=ratio)
    colors = synth.mean_colors(img, pts, tris, ratio) if flavour else None
    slot = capi.IMAGE_B if flavour else capi.IMAGE_A
    ctx = capi.Context(0, W, H)
    ctx.set_image(capi.IMAGE_A, img); ctx.set_image(capi.IMAGE_B, img)
    ctx.upload(pts, tris, colors)
    p = capi.default_params(flavour)
    digests = []
    def check(tag):
        h = hashlib.sha1()
        for b in (capi.BUF_POINTS, capi.BUF_TENERGY, capi.BUF_GRADIENT, capi.BUF_COLNUM):
            h.update(ctx.retrieve(b).tobytes())
        digests.append((tag, h.hexdigest()[:16]))
    ops = ["short", "long", "frame", "frames3", "readback", "upload", "until", "image", "dp", "piecewise", "long", "short"]
    cur_img, dp = img, None
    for step in range(40):
        op = ops[int(rng.integers(len(ops)))] if step > 2 else ["long", "short", "frame"][step]
        if op == "short":
            ctx.iterate(p, int(rng.integers(4, 40)))
            if rng.random() < 0.7: check("short")     # (read back right behind the call, nothing waited for)
        elif op == "long":
            ctx.iterate(p, int(rng.integers(260, 700)))
            if rng.random() < 0.5: check("long")
        elif op == "frame":
            ctx.iterate(p, 1); check("frame")
        elif op == "frames3":
            ctx.iterate(p, 3)
        elif op == "readback":
            check("read-back")
        elif op == "upload":
            moved = ctx.retrieve(capi.BUF_POINTS); ctx.upload(moved, tris, colors)
        elif op == "until":
            n, tot, rel = ctx.iterate_until(p, int(rng.integers(5, 120)), 0.0, 1.0)
            digests.append(("until", "%d %r" % (n, float(tot)))); check("until")
        elif op == "image":
            cur_img = img2 if cur_img is img else img; ctx.set_image(slot, cur_img)
        elif op == "dp":
            dp = None if dp is not None else 0.02; p.dp = 0.0 if dp is None else dp
        elif op == "piecewise":
            ctx.set_dp(0.0 if dp is None else dp); ctx.accumulate(flavour, slot); ctx.energy(flavour); ctx.shift(p.rate); check("piecewise")
    check("end")
    info = (ctx.info(capi.INFO_PERSIST_LAUNCHES), ctx.info(capi.INFO_PERSIST_ITERS), ctx.info(capi.INFO_PERSIST_FAILURES))
    ctx.close()
    return digests, info

if len(sys.argv) > 1 and sys.argv[1] == "--hog":
    hog(); sys.exit(0)
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
bad = 0
for seed in range(1, seeds + 1):
    flavour = seed & 1
    t0 = time.time(); ref, info0 = sequence(seed, flavour); t_alone = time.time() - t0
    hogs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--hog", str(14.0 + 2.0 * t_alone)], stdout=subprocess.PIPE, text=True) for _ in range(1)]
    time.sleep(10.0)   # (the hogs' start-up: import, raster, tables, plan)
    t0 = time.time(); got, info1 = sequence(seed, flavour); t_cont = time.time() - t0
    out = []
    for h in hogs:
        try:
            out.append(h.communicate(timeout=120)[0].strip())
        except Exception:  # noqa: BLE001
            h.kill(); out.append("hog killed")
    print("   %.1f s alone, %.1f s contended; %s" % (t_alone, t_cont, "; ".join(out)), flush=True)
    same = ref == got
    first = next((i for i, (a, b) in enumerate(zip(ref, got)) if a != b), None)
    print("seed %d flavour %d: %d checks, alone launches/iters/given up %s, contended %s: %s%s" %
          (seed, flavour, len(ref), info0, info1, "the same digests" if same else "DIFFERENT", "" if same else " from check %s %s" % (first, ref[first][0] if first is not None else "?")), flush=True)
    bad += 0 if same else 1
print("contention soak: %d seeds, %d differences" % (seeds, bad))
sys.exit(1 if bad else 0)
