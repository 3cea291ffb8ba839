import ctypes, os, sys
ROOT='.'
sys.path.insert(0, ROOT)
os.environ["TPOSE_HIP_LIB"] = os.path.join(os.getcwd(), "tpose_amd", "variants", "libtpose_hip_debugwaves.so")
import numpy as np
from tpose_amd import capi, synth
contrast=float(os.environ.get("TPOSE_CONTRAST","0.3"))
from tpose_amd import photos
img, pts, tris, he, ratio, raster_label = photos.raster_from_env(2048, 2048, 3000, 0.3)
ctx = capi.Context(0, 2048, 2048); ctx.set_image(capi.IMAGE_A, img); ctx.upload(pts, tris, None)
p = capi.default_params(0)
if os.environ.get("TPOSE_TIMELINE_PREPARE"): ctx.prepare(p)   # (the bench's shape: the plan with the probe's speeds)
age = int(os.environ.get("TPOSE_AGE", "1024"))
while age > 0:
    ctx.iterate(p, min(age, 4096)); age -= 4096
ctx.iterate(p, 130); ctx.synchronize()
lib = ctx.lib; lib.tp_debug_dump_persist.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
parts = ctx.info(capi.INFO_PATCHES)
IT, WIT, WAVES = 64, 32, 12
base = 512 * IT * 16
buf = np.zeros(base + 512 * WIT * 16 * 16, np.uint64)
assert lib.tp_debug_dump_persist(ctx.h, buf.ctypes.data, buf.size) == 0
st = buf[base: base + parts * WIT * WAVES * 16].reshape(parts, WIT, WAVES, 16).astype(np.int64)
# per WG: P3 duration = max over waves of (stamp6 'folded') - P1 barrier (stamp4)
dur = (st[:, 8:, :, 6].max(axis=2) - st[:, 8:, :, 4].min(axis=2)) / 100.0
med = np.median(dur, axis=1)
order = np.argsort(-med)
names = {4: "P1 barrier", 14: "LDS pass", 5: "pass", 15: "sums", 6: "folded", 7: "P3 barrier"}
for b in list(order[:6]) + [order[len(order) // 2], order[-1]]:
    it = 12
    t0 = st[b, it, :, 4].min()
    print("block", b, "median P3 %.2f" % med[b])
    for k in (4, 14, 5, 15, 6, 7):
        print("   %-10s" % names[k], " ".join("%5.2f" % ((st[b, it, w, k] - t0) / 100.0) for w in range(WAVES)))
    note = st[b, it, :, 9]
    print("   lanes with more rows than records", " ".join("%d" % (v & 0xffff) for v in note), "| lane-items without a slot: in LDS %d, beyond %d" % (note[0] >> 32, (note[0] >> 16) & 0xffff))
