#!/usr/bin/env python
"""Where the host's share of a short call goes: per-call wall time of tp_iterate (enqueue), tp_synchronize (spin on the pinned word) and
torch.cuda.synchronize() behind it, for calls of 20 grad-iters at the metric size; and the same region without torch's synchronise."""
import sys, time
sys.path.insert(0, '.')
import torch
from tpose_amd import capi, synth
W = H = 2048; NT = 3000
img, pts, tris, he, ratio = synth.workload(W, H, NT, contrast=0.1)
p = capi.default_params(0)
c = capi.Context(0, W, H); c.set_image(capi.IMAGE_A, img); c.upload(pts, tris, None); c.prepare(p); c.iterate(p, 2048); c.synchronize()
torch.cuda.synchronize()
N = 200
ta = tb = tc = 0.0
for _ in range(N):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); c.iterate(p, 20)
    t1 = time.perf_counter(); c.synchronize()
    t2 = time.perf_counter(); torch.cuda.synchronize()
    t3 = time.perf_counter()
    ta += t1 - t0; tb += t2 - t1; tc += t3 - t2
print("20-step call: tp_iterate returns after %.1f us, tp_synchronize after %.1f more, torch.cuda.synchronize() after %.1f more: %.1f us = %.2f us/step"
      % (ta / N * 1e6, tb / N * 1e6, tc / N * 1e6, (ta + tb + tc) / N * 1e6, (ta + tb + tc) / N * 1e6 / 20))
t = 0.0
for _ in range(N):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); torch.cuda.synchronize(); t += time.perf_counter() - t0
print("torch.cuda.synchronize() on an idle device: %.1f us" % (t / N * 1e6))
c.timer_start(); c.iterate(p, 20); d = c.timer_stop()
ds = []
for _ in range(50):
    c.timer_start(); c.iterate(p, 20); ds.append(c.timer_stop())
print("HIP events around a 20-step call on the library's stream: median %.1f us" % sorted(ds)[len(ds) // 2])
