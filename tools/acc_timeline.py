#!/usr/bin/env python
"""In-kernel timeline of k_accumulate at the headline workload (debug build knob TPOSE_DEBUG_ACC=8:
thread 0 of every workgroup stamps wall_clock64, 100 MHz, at its phase boundaries).  Needs an MI355X.
Prints one JSON object: when workgroups start (dispatch ramp), how long each phase takes, when they end."""
import ctypes
import json
import os
import sys

os.environ["TPOSE_DEBUG_ACC"] = "8"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from tpose_amd import capi, synth  # noqa: E402

W = H = 2048
NB = 768  # resident workgroups of k_accumulate (accumulate_grid)
NT = 3000
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
    st = buf.reshape(NB, 16).astype(np.int64)
    acc.append(st)
st = np.stack(acc)  # [rep, block, stamp]
t0 = st[:, :, 0].min(axis=1, keepdims=True)
names = ["start", "t0_prefix_done", "t0_barrier", "t0_walk_done", "t0_barrier2", "t1_prefix_done", "t1_barrier",
         "t1_walk_done", "end"]
out = {"units": "us after the first workgroup's first stamp; 32 launches x %d workgroups" % NB, "stamps": {}}
for k, n in enumerate(names):
    v = (st[:, :, k] - t0) / 100.0
    out["stamps"][n] = {"min": float(v.min(axis=1).mean()), "median": float(np.median(v, axis=1).mean()),
                        "p90": float(np.percentile(v, 90, axis=1).mean()), "max": float(v.max(axis=1).mean())}
seg = {}
for a, b in zip(range(0, 8), range(1, 9)):
    d = (st[:, :, b] - st[:, :, a]) / 100.0
    seg[names[a] + " -> " + names[b]] = {"mean": float(d.mean()), "p90": float(np.percentile(d, 90))}
out["segments"] = seg
start = (st[:, :, 0] - t0) / 100.0
out["start_by_xcd_mean"] = [float(start[:, x::8].mean()) for x in range(8)]
out["start_by_dispatch_order_in_xcd"] = [float(start[:, x::8][:, k::16].mean()) for x in (0,) for k in range(0, 16)]
print(json.dumps(out, indent=1))

# distribution of whole-block times and of the two walks (barrier to barrier: all waves done)
blk = (st[:, :, 8] - st[:, :, 0]) / 100.0
w0 = (st[:, :, 4] - st[:, :, 2]) / 100.0
w1 = (st[:, :, 8] - st[:, :, 6]) / 100.0
p1a = (st[:, :, 2] - st[:, :, 0]) / 100.0
p1b = (st[:, :, 6] - st[:, :, 4]) / 100.0
def pct(v):
    return {q: float(np.percentile(v, q)) for q in (1, 10, 50, 90, 99, 100)}
extra = {"block_total": pct(blk), "walk0_all_waves": pct(w0), "walk1_all_waves": pct(w1), "fetch+prefix0": pct(p1a),
         "prefix1": pct(p1b),
         "corr_blocktime_same_block_across_launches": float(np.corrcoef(blk[0], blk[-1])[0, 1]),
         "slowest_blocks_launch0": [int(b) for b in np.argsort(-blk[0])[:16]],
         "slowest_blocks_launch31": [int(b) for b in np.argsort(-blk[-1])[:16]]}
print(json.dumps(extra, indent=1))

# list lengths per tile vs the walk times of the workgroup that owns it
cnt = np.zeros(1024, np.int32)
lib.tp_debug_tilecount.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
lib.tp_debug_tilecount(ctx.h, cnt.ctypes.data, 1024)
n0, n1 = cnt[:512], cnt[512:]
print(json.dumps({"nlist": pct(cnt), "corr_walk0_nlist": float(np.corrcoef(w0.mean(axis=0), n0)[0, 1]),
                  "corr_walk1_nlist": float(np.corrcoef(w1.mean(axis=0), n1)[0, 1]),
                  "walk0_mean_blocks_lt256": float(w0[:, :256].mean()), "walk0_mean_blocks_ge256": float(w0[:, 256:].mean()),
                  "walk0_by_nlist": {int(k): float(w0.mean(axis=0)[n0 == k].mean()) for k in np.unique(n0)},
                  "walk1_by_nlist": {int(k): float(w1.mean(axis=0)[n1 == k].mean()) for k in np.unique(n1)},
                  "blocktime_by_sum": {int(k): float(blk.mean(axis=0)[(n0 + n1) == k].mean()) for k in np.unique(n0 + n1)}}))

# inside the walk (wave 0, its first work item): set-up done, rows done, static-table part done
def seg(a, b):
    d = (st[:, :, b] - st[:, :, a]) / 100.0
    ok = (st[:, :, a] > 0) & (st[:, :, b] > 0) & (d >= 0) & (d < 50)
    return float(d[ok].mean()) if ok.any() else None
print(json.dumps({"walk0: barrier -> setup done": seg(2, 9), "walk0: setup -> rows done": seg(9, 10), "walk0: rows -> table part done": seg(10, 11),
                  "walk0: table part -> walk_done stamp": seg(11, 3),
                  "walk1: barrier -> setup done": seg(6, 12), "walk1: setup -> rows done": seg(12, 13), "walk1: rows -> table part done": seg(13, 14),
                  "walk1: table part -> walk_done stamp": seg(14, 7)}))
