#!/usr/bin/env python
"""What the plan of an AGED mesh looks like: run the metric workload N grad-iters on the GPU (default 131072), then cut the plan for the positions it
ends with on the CPU (the emulator's build of tp_plan.h) and print, per patch, the rows per lane and the lane-items beyond the 768 whose records
the threads keep.
python tools/aged_plan.py [N] [contrast]"""
import sys, os, subprocess, ctypes as C
import numpy as np
sys.path.insert(0, '.')
from tpose_amd import capi, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
contrast = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
W = H = 2048
os.environ.setdefault("TPOSE_CONTRAST", str(contrast))
from tpose_amd import photos
img, pts, tris, he, ratio, raster_label = photos.raster_from_env(W, H, 3000)
print(raster_label)
so = '/tmp/libtp_emul_persist_aged.so'
subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-o", so, "tests/emul/emul_persist.cpp"])
emp = C.CDLL(so)
NP, NT = pts.shape[0], tris.shape[0]


def stats(p, label):
    p = np.ascontiguousarray(p, np.float32)
    out = np.zeros((256, 8), np.int32)
    n = emp.emul_plan_patches(p.ctypes.data_as(C.c_void_p), NP, tris.ctypes.data_as(C.c_void_p), NT, W, H, C.c_float(ratio), C.c_float(10.0), 256, 160 * 1024,
                              out.ctypes.data_as(C.c_void_p), 256)
    o = out[:n]
    beyond = np.maximum(o[:, 4] - 768, 0)
    print("%s: patches %d | rows per lane %s | lane-items min %d mean %.0f max %d | patches with lane-items beyond 768: %d, beyond: mean %.0f max %d, sum %d | rows (lane-items x rows per lane) mean %.0f max %d"
          % (label, n, np.bincount(o[:, 6]).tolist(), o[:, 4].min(), o[:, 4].mean(), o[:, 4].max(), int((beyond > 0).sum()), beyond.mean(), beyond.max(), beyond.sum(),
             (o[:, 4] * o[:, 6]).mean(), (o[:, 4] * o[:, 6]).max()), flush=True)
    print("   beyond, sorted: %s" % np.sort(beyond)[::-1][:40].tolist(), flush=True)


stats(pts, "fresh")
c = capi.Context(0, W, H); c.set_image(capi.IMAGE_A, img); c.upload(pts, tris, None)
p = capi.default_params(0); c.prepare(p)
done = 0
for mark in (4096, 16384, 65536, N):
    while done < mark:
        c.iterate(p, 4096); done += 4096
    c.synchronize()
    q = c.retrieve(capi.BUF_POINTS)
    stats(q, "after %d grad-iters" % done)
