#!/usr/bin/env python
"""The first grad-iters of a persistent launch, one by one (debug flavour): median period and phases of grad-iters 0..23 -- what a short
call (the driver's --steps 20) pays before the launch reaches its steady state."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["TPOSE_HIP_LIB"] = os.path.join(ROOT, "tpose_amd", "variants", "libtpose_hip_debug.so")
import numpy as np
from tpose_amd import capi, synth
from tpose_amd import photos
img, pts, tris, he, ratio, label = photos.raster_from_env(2048, 2048, 3000)   # (TPOSE_PHOTO=meninas: the headline picture)
print(label)
ctx = capi.Context(0, 2048, 2048); ctx.set_image(capi.IMAGE_A, img); ctx.upload(pts, tris, None)
p = capi.default_params(0); ctx.prepare(p)
# TPOSE_LAUNCH_AFTER=n: n grad-iters before the stamped launch (default 300; the driver's shape: 5, then calls of 20)
for n in [int(v) for v in os.environ.get("TPOSE_LAUNCH_AFTER", "300").split(",")]:
    ctx.iterate(p, n)
ctx.iterate(p, 40); ctx.synchronize()
lib = ctx.lib; lib.tp_debug_dump_persist.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
buf = np.zeros(512 * 64 * 16, np.uint64); assert lib.tp_debug_dump_persist(ctx.h, buf.ctypes.data, buf.size) == 0
st = buf.reshape(512, 64, 16)[:256, :40].astype(np.int64)
t0 = st[:, 0, 0].min()
print("workgroups enter grad-iter 0 within %.1f us of each other" % ((st[:, 0, 0].max() - t0) / 100.0))
e = st[:, 0, 12]
if (e > 0).all():
    print("kernel entry: workgroups within %.1f us of each other; tables in LDS %.2f us after entry (median), prologue over %.2f, first grad-iter begins %.2f; last workgroup's first grad-iter begins %.1f us after the first workgroup's entry"
          % ((e.max() - e.min()) / 100.0, np.median(st[:, 0, 13] - e) / 100.0, np.median(st[:, 0, 14] - e) / 100.0, np.median(st[:, 0, 0] - e) / 100.0, (st[:, 0, 0].max() - e.min()) / 100.0))
if (st[:, 0, 6] > 0).all() and (st[:, 0, 7] > 0).all():
    print("grad-iter 0, P1 in parts (us, medians): line set-up + barrier %.2f, the cut of the lines on one wave + barrier %.2f, lane-item table + the lanes' items + barrier %.2f"
          % (np.median(st[:, 0, 6] - st[:, 0, 1]) / 100.0, np.median(st[:, 0, 7] - st[:, 0, 6]) / 100.0, np.median(st[:, 0, 2] - st[:, 0, 7]) / 100.0))
print("it  start(med, us after the first workgroup's first stamp)  period  P0  P1  P3  P6 (medians over workgroups) | P1 max, P3 max | P3 of thread 0 in parts: scan + fetch, sums + atomics, barrier (medians)")
for it in range(24):
    s = np.median(st[:, it, 0] - t0) / 100.0
    per = np.median(st[:, it + 1, 0] - st[:, it, 0]) / 100.0
    ph = [np.median(st[:, it, k + 1] - st[:, it, k]) / 100.0 for k in range(4)]
    sub = [np.median(st[:, it, b] - st[:, it, a]) / 100.0 for a, b in ((2, 8), (8, 9), (9, 3))]
    print("%2d  %7.1f  %6.2f  %5.2f %5.2f %5.2f %5.2f | %5.2f %5.2f | %5.2f %5.2f %5.2f" % (it, s, per, ph[0], ph[1], ph[2], ph[3], (st[:, it, 2] - st[:, it, 1]).max() / 100.0, (st[:, it, 3] - st[:, it, 2]).max() / 100.0, sub[0], sub[1], sub[2]))
print("grad-iters 0..19 end %.1f us after the first stamp (steady state would be %.1f)" % (np.median(st[:, 20, 0] - t0) / 100.0, 20 * np.median(st[:, 30:39, 0][:, 1:] - st[:, 30:39, 0][:, :-1]) / 100.0))

# the cuts of the lines inside the stamped window: set-up (stamp 1 -> 6), pass A on one wave (6 -> 7), passes B-D (7 -> 2)
for it in range(1, 40):
    if (st[:, it, 7] > 0).all() and np.median(st[:, it, 2] - st[:, it, 1]) > 250:
        print("cut in grad-iter %d: set-up %.2f us, pass A (what every line wants, one wave) %.2f, passes B-D (release, allocate, take + list) %.2f"
              % (it, np.median(st[:, it, 6] - st[:, it, 1]) / 100.0, np.median(st[:, it, 7] - st[:, it, 6]) / 100.0, np.median(st[:, it, 2] - st[:, it, 7]) / 100.0))
