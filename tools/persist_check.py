"""Persistent grad-iter kernel against the two-kernel path and the oracle: parity on small cases, timing at the
metric size.  Run on the GPU box: python tools/persist_check.py [--quick]"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np

from oracle import oracle as O
from tpose_amd import capi, synth
from util import RATE, case


def parity(W, H, grid, flavour, iters):
    img, imgB, pts, tris, ratio, colors = case(W, H, grid)
    sweep = imgB if flavour else img
    out = {}
    for mode in (0, 1):
        ctx = capi.Context(0, W, H)
        ctx.set_image(capi.IMAGE_A, img)
        ctx.set_image(capi.IMAGE_B, imgB)
        ctx.upload(pts, tris, colors if flavour else None)
        ctx.set_persistent(mode)
        ctx.iterate(capi.default_params(flavour), iters)
        out[mode] = (ctx.retrieve(capi.BUF_POINTS), ctx.retrieve(capi.BUF_TENERGY), ctx.retrieve(capi.BUF_GRADIENT),
                     ctx.info(capi.INFO_PATCHES), ctx.info(capi.INFO_PERSIST_ITERS), ctx.info(capi.INFO_CENSUS))
        ctx.close()
    ref = O.iterate(sweep, pts, tris, flavour, ratio, RATE[flavour], iters, colors=colors if flavour else None, literal=False)
    okp = np.array_equal(out[1][0].view(np.uint32), ref["points"].view(np.uint32))
    oke = np.array_equal(out[1][1], ref["ten"]) and np.array_equal(out[1][2], ref["gr"])
    ok2 = np.array_equal(out[0][0].view(np.uint32), ref["points"].view(np.uint32))
    nbad = int((out[1][0].view(np.uint32) != ref["points"].view(np.uint32)).any(axis=1).sum())
    print("parity %dx%d grid %s flavour %d iters %d: patches %d persist_iters %d census %d | persistent points %s energies %s | "
          "two-kernel points %s | vertices off %d / %d" % (W, H, grid, flavour, iters, out[1][3], out[1][4], out[1][5], okp, oke, ok2, nbad, pts.shape[0]), flush=True)
    return okp and oke


def timing(W, H, NT, steps, flavour=0, contrast=0.1):
    img, pts, tris, he, ratio = synth.workload(W, H, NT, contrast=contrast)
    res = {}
    for mode in (1, 0):
        ctx = capi.Context(0, W, H)
        ctx.set_image(capi.IMAGE_A, img)
        ctx.set_image(capi.IMAGE_B, img)
        colors = synth.mean_colors(img, pts, tris, ratio) if flavour else None
        ctx.upload(pts, tris, colors)
        ctx.set_persistent(mode)
        p = capi.default_params(flavour)
        ctx.prepare(p)
        ctx.iterate(p, 64)
        ctx.synchronize()
        ts = []
        for rep in range(3):
            t0 = time.perf_counter()
            ctx.iterate(p, steps)
            ctx.synchronize()
            ts.append((time.perf_counter() - t0) / steps * 1e6)
        res[mode] = (min(ts), ctx.retrieve(capi.BUF_POINTS), ctx.info(capi.INFO_PATCHES), ctx.info(capi.INFO_PATCH_LDS), ctx.info(capi.INFO_PATCH_LINES))
        ctx.close()
    same = np.array_equal(res[0][1].view(np.uint32), res[1][1].view(np.uint32))
    print("timing %dx%d / %d triangles contrast %g flavour %d, %d steps: persistent %.2f us/iter (patches %d, LDS %d B, lines walked %d) | "
          "two-kernel %.2f us/iter | same positions after %d iters: %s" % (W, H, tris.shape[0], contrast, flavour, steps, res[1][0], res[1][2], res[1][3], res[1][4],
                                                                    res[0][0], 64 + 3 * steps, same), flush=True)
    return same


if __name__ == "__main__":
    quick = "--quick" in sys.argv
    ok = True
    for (W, H, grid) in [(64, 48, (6, 4)), (300, 200, (15, 5)), (128, 32, None), (640, 480, (50, 30))]:
        for fl in (0, 1):
            ok &= parity(W, H, grid, fl, 6)
    ok &= parity(300, 200, (15, 5), 0, 200)
    ok &= timing(2048, 2048, 3000, 128)
    ok &= timing(2048, 2048, 3000, 512)
    ok &= timing(2048, 2048, 3000, 2048)
    if not quick:
        ok &= timing(2048, 2048, 3000, 2048, flavour=1)
        ok &= timing(4096, 4096, 12000, 512)
        ok &= timing(674, 449, 150, 512)
    print("ALL OK" if ok else "FAILURES", flush=True)
    sys.exit(0 if ok else 1)
