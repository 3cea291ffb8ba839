#!/usr/bin/env python
"""Time the fused grad-iter on the BASELINE.json configurations other than the headline one
(profiles/README.md quotes the output).  Needs an MI355X."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from tpose_amd import capi, synth  # noqa: E402


def run(name, W, H, NT, flavour, steps):
    img, pts, tris, he, ratio = synth.workload(W, H, NT)
    ctx = capi.Context(0, W, H)
    ctx.set_image(capi.IMAGE_A, img)
    colors = None
    if flavour == capi.WARP:
        ctx.set_image(capi.IMAGE_B, synth.displaced_raster(img))
        colors = synth.mean_colors(img, pts, tris, ratio)
    ctx.upload(pts, tris, colors)
    p = capi.default_params(flavour)
    ctx.iterate(p, 64)
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.iterate(p, steps)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    acc = ctx.profile_iterate(p, 64)
    bytes_iter = 4 * W * H + 16 * NT + 24 * 13 * NT + 24 * pts.shape[0]
    out = dict(config=name, raster=[W, H], triangles=NT, flavour="warp" if flavour else "triangulate",
               us_per_iter=dt / steps * 1e6, tri_iters_per_s=NT * steps / dt, k_lines_us=acc,
               roofline_frac=bytes_iter / (acc * 1e-6) / 8e12)
    ctx.close()
    print(json.dumps(out), flush=True)


def run_readback(W, H, NT, steps):
    """the reference's frame: one grad-iter, then four blocking readbacks
    (software/triangulate/main.cpp:196-204: terr, perr, cn, points)"""
    img, pts, tris, he, ratio = synth.workload(W, H, NT)
    ctx = capi.Context(0, W, H)
    ctx.set_image(capi.IMAGE_A, img)
    ctx.upload(pts, tris, None)
    p = capi.default_params(capi.TRIANGULATE)
    for warm in range(2):
        t0 = time.perf_counter()
        for k in range(steps):
            ctx.iterate(p, 1)
            ctx.retrieve(capi.BUF_TENERGY); ctx.retrieve(capi.BUF_PENERGY)
            ctx.retrieve(capi.BUF_COLNUM); ctx.retrieve(capi.BUF_POINTS)
        dt = time.perf_counter() - t0
    for warm in range(2):
        t0 = time.perf_counter()
        for k in range(steps):
            ctx.iterate(p, 1)
            ctx.retrieve_many([capi.BUF_TENERGY, capi.BUF_PENERGY, capi.BUF_COLNUM, capi.BUF_POINTS])
        dt2 = time.perf_counter() - t0
    ctx.close()
    print(json.dumps(dict(config="readback every iter %dx%d / %d" % (W, H, NT), us_per_iter_four_calls=dt / steps * 1e6,
                          us_per_iter_one_batched_call=dt2 / steps * 1e6, tri_iters_per_s=NT * steps / dt2)), flush=True)


def run_cold(W, H, NT, nctx, reps):
    """cold-cache sweep: cycle through nctx contexts, each with its own plane, so that every
    k_lines launch finds its table neither in L2 nor in the 256 MB Infinity Cache"""
    ctxs = []
    for k in range(nctx):
        img, pts, tris, he, ratio = synth.workload(W, H, NT, seed=1234 + k)
        c = capi.Context(0, W, H)
        c.set_image(capi.IMAGE_A, img)
        c.upload(pts, tris, None)
        ctxs.append(c)
    p = capi.default_params(capi.TRIANGULATE)
    for c in ctxs:
        c.profile_iterate(p, 2)
    tot, n = 0.0, 0
    for r in range(reps):
        for c in ctxs:
            tot += c.profile_iterate(p, 1)
            n += 1
    warm = ctxs[0].profile_iterate(p, 64)
    bytes_iter = 4 * W * H + 16 * NT + 24 * 13 * NT + 24 * ctxs[0].NP
    print(json.dumps(dict(config="cold cache %dx%d / %d, %d prefix tables = %.0f MB cycled" % (W, H, NT, nctx, nctx * 8 * W * H / 1e6),
                          k_lines_us_cold=tot / n, k_lines_us_warm_same_process=warm,
                          roofline_frac_cold=bytes_iter / (tot / n * 1e-6) / 8e12)), flush=True)
    for c in ctxs:
        c.close()


def run_concurrent(W, H, NT, nctx, steps):
    """nctx independent problems on ONE GPU, each in its own context and stream (the batch configuration has more
    pairs than GPUs to spare): the single-problem path is latency-bound, so they overlap"""
    ctxs = []
    for k in range(nctx):
        img, pts, tris, he, ratio = synth.workload(W, H, NT, seed=1234 + k)
        c = capi.Context(0, W, H)
        c.set_image(capi.IMAGE_A, img)
        c.upload(pts, tris, None)
        ctxs.append(c)
    p = capi.default_params(capi.TRIANGULATE)
    for c in ctxs:
        c.iterate(p, 64)
    for c in ctxs:
        c.synchronize()
    t0 = time.perf_counter()
    for start in range(0, steps, 256):  # interleave the enqueues so that no stream runs dry
        for c in ctxs:
            c.iterate(p, 256)
    for c in ctxs:
        c.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps(dict(config="%d concurrent contexts on one GPU, %dx%d / %d each" % (nctx, W, H, NT),
                          aggregate_tri_iters_per_s=nctx * NT * steps / dt, us_per_iter_per_context=dt / steps * 1e6)), flush=True)
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    run("headline 2048^2 / 3000", 2048, 2048, 3000, capi.TRIANGULATE, 2048)
    run("warp flavour 2048^2 / 3000", 2048, 2048, 3000, capi.WARP, 2048)
    run("batch element 4096^2 / 12000", 4096, 4096, 12000, capi.TRIANGULATE, 1024)
    run("plumbing size 674x449 / 150", 674, 449, 150, capi.TRIANGULATE, 4096)
    run("start state 2048^2 / 2", 2048, 2048, 2, capi.TRIANGULATE, 1024)
    run_readback(2048, 2048, 3000, 512)
    run_cold(2048, 2048, 3000, 18, 4)
    run_cold(4096, 4096, 12000, 5, 8)
    run_concurrent(2048, 2048, 3000, 2, 2048)
    run_concurrent(2048, 2048, 3000, 4, 2048)
