#!/usr/bin/env python
"""What the planner's weights are worth: per patch, the stale rows k_persist really fetched over one launch (counting flavour of the library,
-DPK_DBG_STALE, built here into tpose_amd/variants/) beside what the planner weighed the patch with.  Needs an MI355X.
TPOSE_PHOTO / TPOSE_CONTRAST as everywhere; TPOSE_COUNT_AFTER=n grad-iters first (default 5: the driver's warm-up), TPOSE_COUNT_ITERS (default 20).
Prints one JSON object; per-block arrays (index = blockIdx.x, as in tools/persist_timeline.py's TPOSE_TIMELINE_DUMP)."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib_path = os.path.join(ROOT, "tpose_amd", "variants", "libtpose_hip_count.so")
os.environ["TPOSE_HIP_LIB"] = lib_path
from tpose_amd import build as tb  # noqa: E402
if not os.path.exists(lib_path) or "--rebuild" in sys.argv:
    os.makedirs(os.path.dirname(lib_path), exist_ok=True)
    tb.build(force=True, extra=["-DPK_DBG_STALE"], out=lib_path)
import numpy as np  # noqa: E402
from tpose_amd import capi, photos  # noqa: E402
W = H = 2048; NT = 3000
img, pts, tris, he, ratio, label = photos.raster_from_env(W, H, NT)
c = capi.Context(0, W, H); c.set_image(capi.IMAGE_A, img); c.upload(pts, tris, None)
p = capi.default_params(0); c.prepare(p)
after = int(os.environ.get("TPOSE_COUNT_AFTER", "5")); n = int(os.environ.get("TPOSE_COUNT_ITERS", "20"))
if after: c.iterate(p, after)
lib = c.lib
cnt = np.zeros((512, 4), np.uint64)
lib.tp_debug_persist_counts.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
lib.tp_debug_plan_weights.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
assert lib.tp_debug_persist_counts(c.h, cnt.ctypes.data, 1) == 0
lib.tp_debug_persist_vcounts.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
lib.tp_debug_plan_owner.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
vc = np.zeros((pts.shape[0], 2), np.uint64)
assert lib.tp_debug_persist_vcounts(c.h, vc.ctypes.data, 1) == 0
pts0 = c.retrieve(capi.BUF_POINTS).copy()
c.iterate(p, n); c.synchronize()
wts = np.zeros((512, 4), np.float32)
rc = lib.tp_debug_plan_weights(c.h, wts.ctypes.data, wts.size)   # (the plan the launch ran on, unless it was cut again inside the call)
assert lib.tp_debug_persist_counts(c.h, cnt.ctypes.data, 1) == 0
assert lib.tp_debug_persist_vcounts(c.h, vc.ctypes.data, 1) == 0
if os.environ.get("TPOSE_COUNT_DUMP"):   # for tools that try other weights on the same mesh: positions at the window's start and end, what every vertex's lines fetched and walked, its patch
    owner = np.zeros(pts.shape[0], np.int32)
    lib.tp_debug_plan_owner(c.h, owner.ctypes.data, owner.size)
    np.savez(os.environ["TPOSE_COUNT_DUMP"], pts0=pts0, pts1=c.retrieve(capi.BUF_POINTS), tris=tris, vstale=vc[:, 0] / float(n), vrows=vc[:, 1] / float(n), owner=owner, ratio=ratio, W=W, H=H, iters=n)
parts = c.info(capi.INFO_PATCHES)
blk = np.arange(parts); patch = (blk & 7) * (parts >> 3) + (blk >> 3) if parts % 8 == 0 else blk
out = {"workload": "%dx%d / %d triangles, %s; grad-iters %d..%d" % (W, H, NT, label, after, after + n - 1), "patches": parts, "plan_weights_rc": rc,
       "per grad-iter, by block": {"stale rows": (cnt[:parts, 0] / n).round(1).tolist(), "wave-loads": (cnt[:parts, 1] / n).round(1).tolist(),
                                   "lane-items moved": (cnt[:parts, 2] / n).round(1).tolist(), "rows": (cnt[:parts, 3] / n).round(1).tolist()},
       "planner, by block": {"work": wts[patch, 0].round(1).tolist(), "rows": wts[patch, 1].round(1).tolist(), "hot": wts[patch, 2].astype(int).tolist(), "rows per lane": wts[patch, 3].astype(int).tolist()}}
print(json.dumps(out))
