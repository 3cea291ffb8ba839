import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from tpose_amd import capi, synth
img, pts, tris, he, ratio = synth.workload(800, 920, 3000) if False else synth.workload(2048, 2048, 3000)
ctx = capi.Context(0, 2048, 2048); ctx.set_image(capi.IMAGE_A, img)
ctx.upload(pts, tris, None)
for rep in range(3):
    t0 = time.perf_counter()
    for k in range(200): ctx.upload(pts, tris, None)
    print("upload NT=3000: %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
t0 = time.perf_counter()
for k in range(200): ctx.accumulate(0, 0)
print("accumulate (piecewise, with its wait): %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
t0 = time.perf_counter()
for k in range(200): ctx.energy(0); ctx.retrieve(capi.BUF_TENERGY)
print("energy + retrieve tenergy: %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
