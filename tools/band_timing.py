"""Row e3 (one direction of a pair over two GPUs), what the one-GPU box can say about it: the band split's protocol at the
4096^2 / 12 000-triangle size with both bands on device 0 (two contexts, two streams, mailboxes in device memory).
  * one context, 256 patches                          -- the unsplit direction
  * two bands x 128 patches                           -- the same 256 workgroups, hand-overs through the bands' mailboxes
    (system-scope stores into both, final positions collected per launch): what the protocol itself costs
  * two bands x 64 patches                            -- half the workgroups: how a grad-iter scales with the patches it is cut into
On two GPUs every band has a whole device (2 x 256 patches); the link's latency is what this box cannot show.
Prints one JSON object.  Needs an MI355X."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

from tpose_amd import capi, synth

W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
NT = int(sys.argv[2]) if len(sys.argv) > 2 else 12000
STEPS = 512
img, pts, tris, he, ratio = synth.workload(W, H, NT, contrast=0.1)
p = capi.default_params(0)


def make(n_bands, patches):
    cap_p, cap_t = pts.shape[0] + 64, tris.shape[0] + 64
    nbytes = capi.band_mailbox_bytes(cap_p, cap_t)
    boxes = [torch.zeros(nbytes // 8 + 1, dtype=torch.int64, device="cuda:0") for _ in range(n_bands)]
    torch.cuda.synchronize()
    ctxs = []
    for b in range(n_bands):
        ctx = capi.Context(0, W, H)
        ctx.set_image(capi.IMAGE_A, img)
        ctx.upload(pts, tris, None)
        if n_bands > 1:
            ctx.band_attach(b, n_bands, [bx.data_ptr() for bx in boxes], nbytes, cap_p, cap_t, patches)
        ctx.prepare(p)
        ctx.synchronize()
        ctxs.append(ctx)
    return ctxs, boxes


def run(n_bands, patches):
    ctxs, boxes = make(n_bands, patches)
    for c in ctxs:
        c.iterate(p, 64)
    for c in ctxs:
        c.synchronize()
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        for c in ctxs:
            c.iterate(p, STEPS)
        for c in ctxs:
            c.synchronize()
        ts.append((time.perf_counter() - t0) / STEPS * 1e6)
    out = {"us_per_grad_iter": round(min(ts), 2), "patches_of_the_plan": ctxs[0].info(capi.INFO_PATCHES),
           "launches_given_up": sum(c.info(9) for c in ctxs), "persistent_grad_iters": ctxs[0].info(capi.INFO_PERSIST_ITERS)}
    pos = [c.retrieve(capi.BUF_POINTS) for c in ctxs]
    for c in ctxs:
        c.close()
    return out, pos[0]


res = {"workload": "%dx%d / %d triangles, contrast 0.1, triangulate flavour, %d grad-iters per call, both bands on device 0" % (W, H, tris.shape[0], STEPS)}
res["one context, 256 patches"], ref = run(1, 0)
for nb, pp in ((2, 128), (2, 64)):
    res["%d bands x %d patches" % (nb, pp)], got = run(nb, pp)
    res["%d bands x %d patches" % (nb, pp)]["same positions as the one context"] = bool(np.array_equal(got.view(np.uint32), ref.view(np.uint32)))
print(json.dumps(res, indent=1))
