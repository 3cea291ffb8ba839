#!/usr/bin/env python
"""In-kernel timeline of k_accumulate at the headline workload (debug knob TPOSE_DEBUG_ACC=8: thread 0 of every
workgroup stamps wall_clock64, 100 MHz, at its phase boundaries; inside the walk, the first work item of wave 0).
Needs an MI355X.  Prints one JSON object: when workgroups start, how long each phase takes, when they end."""
import ctypes
import json
import os
import sys

os.environ["TPOSE_DEBUG_ACC"] = "8"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from tpose_amd import capi, synth  # noqa: E402

W = H = 2048
NT = 3000
NB = 768  # resident workgroups of k_accumulate (3 per CU); 1024 tiles: workgroups < 256 walk two tiles, the rest one
img, pts, tris, he, ratio = synth.workload(W, H, NT)
ctx = capi.Context(0, W, H)
ctx.set_image(capi.IMAGE_A, img)
ctx.upload(pts, tris, None)
p = capi.default_params(capi.TRIANGULATE)
ctx.profile_iterate(p, 4)  # eager launches first: the debug buffer is allocated outside graph capture
ctx.iterate(p, 64)
ctx.synchronize()
lib = ctx.lib
lib.tp_debug_dump.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
acc = []
for rep in range(32):
    ctx.iterate(p, 1)
    ctx.synchronize()
    buf = np.zeros(NB * 16, np.uint64)
    rc = lib.tp_debug_dump(ctx.h, buf.ctypes.data, buf.size)
    assert rc == 0, rc
    acc.append(buf.reshape(NB, 16).astype(np.int64))
st = np.stack(acc)  # [launch, workgroup, stamp]
t0 = st[:, :, 0].min(axis=1, keepdims=True)
two, one = slice(0, 256), slice(256, NB)


def pct(v):
    return {str(q): round(float(np.percentile(v, q)), 2) for q in (1, 50, 90, 100)}


def dur(sel, a, b):
    d = (st[:, sel, b] - st[:, sel, a]) / 100.0
    ok = (st[:, sel, a] > 0) & (st[:, sel, b] > 0) & (d >= 0) & (d < 100)
    return pct(d[ok]) if ok.any() else None


# stamps: 0 start | per tile: prefix done, barrier, walk done (wave 0), barrier | 9..11 / 12..14: inside the walk
out = {
    "units": "us; percentiles over 32 launches x workgroups; times are after the first workgroup's first stamp",
    "start": {"two-tile workgroups": pct((st[:, two, 0] - t0) / 100.0), "one-tile workgroups": pct((st[:, one, 0] - t0) / 100.0)},
    "end": {"two-tile workgroups": pct((st[:, two, 8] - t0) / 100.0), "one-tile workgroups": pct((st[:, one, 4] - t0) / 100.0)},
    "two-tile workgroups": {
        "fetch + prefix 0 (to barrier)": dur(two, 0, 2), "walk 0 (barrier to barrier, all waves)": dur(two, 2, 4),
        "prefix 1": dur(two, 4, 6), "walk 1": dur(two, 6, 8),
        "walk 0: set-up": dur(two, 2, 9), "walk 0: rows": dur(two, 9, 10), "walk 0: static part": dur(two, 10, 11),
    },
    "one-tile workgroups": {
        "fetch + prefix (to barrier)": dur(one, 0, 2), "walk (barrier to barrier, all waves)": dur(one, 2, 4),
        "walk: set-up": dur(one, 2, 9), "walk: rows": dur(one, 9, 10), "walk: static part": dur(one, 10, 11),
    },
}
print(json.dumps(out, indent=1))
