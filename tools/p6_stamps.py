import ctypes, os, sys
sys.path.insert(0, '.')
os.environ["TPOSE_HIP_LIB"] = os.path.join(os.getcwd(), "tpose_amd", "variants", "libtpose_hip_debugwaves.so")
import numpy as np
from tpose_amd import capi, synth
img, pts, tris, he, ratio = synth.workload(2048, 2048, 3000, contrast=0.1)
ctx = capi.Context(0, 2048, 2048); ctx.set_image(capi.IMAGE_A, img); ctx.upload(pts, tris, None)
p = capi.default_params(0); ctx.iterate(p, 1024); ctx.iterate(p, 130); ctx.synchronize()
lib = ctx.lib; lib.tp_debug_dump_persist.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
parts = ctx.info(capi.INFO_PATCHES)
IT, WIT, WAVES = 64, 32, 12
base = 512 * IT * 16
buf = np.zeros(base + 512 * WIT * 16 * 16, np.uint64)
assert lib.tp_debug_dump_persist(ctx.h, buf.ctypes.data, buf.size) == 0
st = buf[base: base + parts * WIT * WAVES * 16].reshape(parts, WIT, WAVES, 16).astype(np.int64)
sel = st[:, 8:, :, :]
names = [(7, 11, "P3 barrier -> moments of the first corner variant"), (11, 12, "-> its energy"), (12, 13, "-> the returning atomic back"), (13, 8, "-> steps taken, posted (P6 done)"), (0, 1, "top -> polled"), (2, 3, "P0 barrier -> P1 done"), (3, 4, "-> P1 barrier")]
for a, b, lab in names:
    d = (sel[:, :, :3, b] - sel[:, :, :3, a]) / 100.0
    ok = (sel[:, :, :3, a] > 0) & (sel[:, :, :3, b] > 0)
    print("%-52s median %.2f  p10 %.2f  p90 %.2f us (waves 0-2)" % (lab, np.median(d[ok]), np.percentile(d[ok], 10), np.percentile(d[ok], 90)))
