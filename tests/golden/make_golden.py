#!/usr/bin/env python
"""Generate the golden fixtures of tests/golden/ (run in the development container).

The reference itself cannot produce vectors for this path (GLSL on OpenGL through an un-vendored engine;
SURVEY.md section 8c), so the fixtures come from the CPU oracle's LITERAL reference form and are only
written after an independent NumPy brute force (tests/brute.py) reproduced every integer of them.
They pin the oracle (and with it the HIP path) against drift: tests/test_golden.py compares the
oracle on CPU and the HIP path on the GPU with these files.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import brute  # noqa: E402
from oracle import oracle as O  # noqa: E402
from util import RATE, case  # noqa: E402

CASES = {
    # name: (W, H, grid, flavour, iterations stored)
    "tri_64x48_48": (64, 48, (6, 4), 0, (1, 10, 50)),
    "tri_97x61_48": (97, 61, (6, 4), 0, (1, 10, 50)),
    "tri_33x17_2": (33, 17, None, 0, (1, 10)),
    "warp_64x48_48": (64, 48, (6, 4), 1, (1, 10, 50)),
    "warp_97x61_48": (97, 61, (6, 4), 1, (1, 10)),
    # the start state after split(0); split(1) (state iv: NT 6 / NP 6), built with the host mirror
    "tri_64x48_split6": (64, 48, "split", 0, (1, 10, 50)),
}


def main():
    for name, (W, H, grid, fl, iters) in CASES.items():
        if grid == "split":
            from tpose_amd import hostlib, synth
            img, imgB, _, _, ratio, _ = case(W, H, None)
            hostlib.set_ratio(ratio)
            t = hostlib.Triangulation()
            assert t.split(0) and t.split(1) and t.NT == 6 and t.NP == 6
            pts, tris = t.points.copy(), t.triangles.copy()
            colors = synth.mean_colors(img, pts, tris, ratio)
        else:
            img, imgB, pts, tris, ratio, colors = case(W, H, grid)
        sweep = imgB if fl else img
        col = colors if fl else None
        # first iteration: every buffer, cross-checked by the brute force
        first = O.iterate(sweep, pts, tris, fl, ratio, RATE[fl], 1, colors=col, literal=True)
        bcn, bca, bten = brute.evaluate(sweep, pts, tris, fl, ratio, colors=colors)
        assert np.array_equal(first["cn"], bcn) and np.array_equal(first["ten"], bten), name
        if fl == 0:
            assert np.array_equal(first["ca"], bca), name
        out = dict(W=W, H=H, flavour=fl, ratio=np.float32(ratio), points=pts, triangles=tris,
                   colors=colors, rate=np.float32(RATE[fl]), seed_note="util.case(W, H, grid)",
                   cn1=first["cn"], ca1=first["ca"], ten1=first["ten"], gr1=first["gr"])
        for n in iters:
            st = O.iterate(sweep, pts, tris, fl, ratio, RATE[fl], n, colors=col, literal=True)
            out["points_%d" % n] = st["points"]
            out["ten_%d" % n] = st["ten"]
        np.savez_compressed(os.path.join(HERE, name + ".npz"), image=sweep, **out)
        print(name, "ok", {k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.ndim})


if __name__ == "__main__":
    main()
