"""Microseconds per grad-iter of calls of 20 / 64 / 256 grad-iters (one launch + one wait each): one context alone, the same
calls with five contexts alive, and five contexts (five rasters, 590 MB of tables: more than the Infinity Cache) in turn.
Runs on the GPU box: python tools/call_length.py"""
import sys, time
sys.path.insert(0, '.')
from tpose_amd import capi, synth
W = H = 2048; NT = 3000
img, pts, tris, he, ratio = synth.workload(W, H, NT, contrast=0.1)
p = capi.default_params(0)
def mk(seed):
    im = synth.workload(W, H, NT, seed=seed, contrast=0.1)[0]
    c = capi.Context(0, W, H); c.set_image(capi.IMAGE_A, im); c.upload(pts, tris, None); c.prepare(p); c.iterate(p, 6144); c.synchronize()
    return c
def loop(cs, steps, rounds=6):
    t0 = time.perf_counter()
    for _ in range(rounds):
        for c in cs:
            c.iterate(p, steps); c.synchronize()
    return (time.perf_counter() - t0) / (rounds * len(cs) * steps) * 1e6
c0 = mk(1)
for steps in (20, 64, 256):
    loop([c0], steps); print("one context alone, %d-step calls: %.2f us/step" % (steps, loop([c0] * 5, steps)), flush=True)
cs = [c0] + [mk(2 + k) for k in range(4)]
for steps in (20, 64, 256):
    loop([c0], steps); print("five alive, calls on one, %d-step calls: %.2f us/step" % (steps, loop([c0] * 5, steps)), flush=True)
    loop(cs, steps); print("five alive, cycled, %d-step calls: %.2f us/step" % (steps, loop(cs, steps)), flush=True)
print("replans", [c.info(capi.INFO_REPLANS) for c in cs]); print("persist iters / failures", [(c.info(capi.INFO_PERSIST_ITERS), c.info(capi.INFO_CENSUS)) for c in cs])
