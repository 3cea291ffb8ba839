import sys, time
sys.path.insert(0, '.')
from tpose_amd import capi, synth
img, pts, tris, he, ratio = synth.workload(4096, 4096, 12000, contrast=0.1)
ctx = capi.Context(0, 4096, 4096)
ctx.set_image(capi.IMAGE_A, img); ctx.upload(pts, tris, None)
p = capi.default_params(0); ctx.prepare(p); ctx.iterate(p, 64); ctx.synchronize()
out = []
for steps in (256, 512, 512, 1024, 1024):
    t0 = time.perf_counter(); ctx.iterate(p, steps); ctx.synchronize()
    out.append("%d: %.2f (rows %d lds %d replans %d)" % (steps, (time.perf_counter() - t0) / steps * 1e6, ctx.info(13), ctx.info(capi.INFO_PATCH_LDS), ctx.info(capi.INFO_REPLANS)))
print(" | ".join(out), "given up", ctx.info(capi.INFO_PERSIST_FAILURES))
