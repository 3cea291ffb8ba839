#!/usr/bin/env python
"""Several host threads, one context each on the same GPU, capturing graphs, uploading, iterating, reading back and
rendering at the same time; every result checked against the oracle.  `python tools/thread_stress.py [rounds] [threads]`"""
import os
import sys
import threading
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402

from oracle import oracle as O  # noqa: E402
from tpose_amd import capi  # noqa: E402
from util import RATE, case  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 4
errors = []


def work(k, rnd):
    try:
        W, H = 320 + 16 * k, 200 + 8 * k
        img, imgB, pts, tris, ratio, colors = case(W, H, (10 + k, 7), seed=100 * rnd + k)
        ctx = capi.Context(0, W, H)
        ctx.set_image(capi.IMAGE_A, img)
        flavour = k % 2
        if flavour:
            ctx.set_image(capi.IMAGE_B, imgB)
        total = 0
        cur = pts
        for rep in range(4):
            ctx.upload(cur, tris, colors if flavour else None)   # fresh upload: new graph capture on the next iterate
            p = capi.default_params(flavour)
            ctx.iterate(p, 16 + rep)
            total += 16 + rep
            cur = ctx.retrieve(capi.BUF_POINTS)
            ctx.render(capi.RENDER_STORED if flavour else capi.RENDER_AVERAGE)
        ten = ctx.retrieve(capi.BUF_TENERGY)
        ctx.close()
        ref = O.iterate(imgB if flavour else img, pts, tris, flavour, ratio, RATE[flavour], total, colors=colors if flavour else None, literal=False)
        if not (np.array_equal(ten, ref["ten"]) and np.array_equal(cur.view(np.uint32), ref["points"].view(np.uint32))):
            errors.append("round %d thread %d: result differs from the oracle" % (rnd, k))
    except Exception:  # noqa: BLE001
        errors.append("round %d thread %d: %s" % (rnd, k, traceback.format_exc()))


for rnd in range(rounds):
    ts = [threading.Thread(target=work, args=(k, rnd)) for k in range(nthreads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
print("thread stress: %d rounds x %d threads, %d errors" % (rounds, nthreads, len(errors)))
for e in errors[:5]:
    print(e)
sys.exit(1 if errors else 0)
