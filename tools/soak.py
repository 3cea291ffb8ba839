#!/usr/bin/env python
"""Randomised parity soak (needs an MI355X; the oracle is the checker): random raster sizes, random triangle
soups and jittered grids, both flavours, random dp, persistent launches on and off, piecewise moments and fused iterations, all
compared bit for bit.  `python tools/soak.py [cases] [seed]`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from oracle import oracle as O  # noqa: E402
from tpose_amd import capi, synth  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
bad = 0
for case in range(cases):
    W, H = int(rng.integers(17, 700)), int(rng.integers(9, 420))
    if case % 9 == 0:
        W, H = int(rng.integers(1000, 2300)), int(rng.integers(600, 1300))
    if case % 11 == 5:
        W, H = int(rng.integers(1, 17)), int(rng.integers(1, 40))  # smaller than one tile
    ratio = float(np.float32(W) / np.float32(H))
    if case % 7 == 3:
        ratio = float(np.float32(rng.choice([0.5, 1.0, 1.7777])))  # tpose::RATIO need not be the raster's aspect
    img = synth.voronoi_raster(W, H, seed=int(rng.integers(1 << 30)), sites=int(rng.integers(3, 40)))
    kind = case % 3
    if kind == 0:  # soup
        NP, NT = int(rng.integers(3, 60)), int(rng.integers(1, 90))
        pts = ((rng.random((NP, 2)).astype(np.float32) * 2 - 1) * np.float32(rng.choice([0.7, 1.0, 1.4])))
        pts[:, 0] *= np.float32(ratio)
        if rng.random() < 0.3:
            pts = (np.round(pts * 16) / 16).astype(np.float32)
        tris = np.zeros((NT, 4), np.int32)
        tris[:, :3] = rng.integers(0, NP, (NT, 3))
    else:
        gx, gy = int(rng.integers(1, 24)), int(rng.integers(1, 16))
        pts, tris, _ = synth.grid_triangulation(gx, gy, ratio=ratio, jitter=float(rng.choice([0.0, 0.25, 0.45])), seed=int(rng.integers(1 << 20)))
    flavour = int(rng.integers(0, 2))
    dp = float(rng.choice([0.0, 0.05, 0.011, 0.2]))  # 0: the reference law
    ctx = capi.Context(0, W, H)
    ctx.set_ratio(ratio)
    ctx.set_image(capi.IMAGE_A, img)
    imgB = synth.displaced_raster(img, amp=5.0) if flavour else None
    colors = None
    if flavour:
        ctx.set_image(capi.IMAGE_B, imgB)
        colors = rng.integers(0, 256, (tris.shape[0], 4)).astype(np.int32)
        colors[:, 3] = 1
    ctx.upload(pts, tris, colors)
    if dp > 0:
        ctx.set_dp(dp)
    dpe = dp if dp > 0 else O.dp(flavour, tris.shape[0])
    sweep = imgB if flavour else img
    # piecewise: exact 64-bit moments
    ctx.accumulate(flavour, capi.IMAGE_B if flavour else capi.IMAGE_A)
    ctx.energy(flavour)
    ok = np.array_equal(ctx.retrieve(capi.BUF_MOMENTS), O.moments(sweep, pts, tris, dpe, ratio))
    # fused iterations (persistent launches or the two-kernel path)
    iters = int(rng.integers(1, 9))
    margin = int(rng.choice([0, 0, 1, 1]))  # persistent launches on / off
    ctx.upload(pts, tris, colors)
    ctx.set_persistent(margin)
    params = capi.default_params(flavour)
    params.dp = dp
    rate = float(np.float32(rng.choice([params.rate, 1e-5, 2e-4])))
    params.rate = rate
    ctx.iterate(params, iters)
    ref = O.iterate(sweep, pts, tris, flavour, ratio, rate, iters, colors=colors, dp_=dpe, literal=False)
    ok &= np.array_equal(ctx.retrieve(capi.BUF_TENERGY), ref["ten"]) and np.array_equal(ctx.retrieve(capi.BUF_COLNUM), ref["cn"])
    ok &= np.array_equal(ctx.retrieve(capi.BUF_POINTS).view(np.uint32), ref["points"].view(np.uint32))
    ok &= np.array_equal(ctx.retrieve(capi.BUF_GRADIENT), ref["gr"].reshape(-1, 2))
    # the piecewise calls in the reference's order give the same state
    ctx.upload(pts, tris, colors)
    ctx.set_persistent(1)
    for k in range(iters):
        ctx.accumulate(flavour, capi.IMAGE_B if flavour else capi.IMAGE_A)
        ctx.energy(flavour)
        ctx.shift(rate)
    ok &= np.array_equal(ctx.retrieve(capi.BUF_POINTS).view(np.uint32), ref["points"].view(np.uint32))
    ok &= np.array_equal(ctx.retrieve(capi.BUF_TENERGY), ref["ten"]) and np.array_equal(ctx.retrieve(capi.BUF_GRADIENT), ref["gr"].reshape(-1, 2))
    ctx.close()
    if not ok:
        bad += 1
        print("MISMATCH case %d: %dx%d kind %d flavour %d dp %g iters %d persistent %d NT %d" % (case, W, H, kind, flavour, dp, iters, margin, tris.shape[0]), flush=True)
print("soak: %d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
