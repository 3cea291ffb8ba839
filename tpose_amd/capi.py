"""ctypes binding of the C ABI in include/tpose_hip.h (libtpose_hip.so).

Thin plumbing for tests, bench.py and the multi-GPU drivers: every call goes straight through the
C entry points a C++/cgo/JNI host would bind.  There is no fallback: a missing library or a
missing GPU raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TPOSE_HIP_LIB") or os.path.join(_HERE, "libtpose_hip.so")  # override: A/B builds

TP_OK, TP_ERR_INVALID, TP_ERR_NO_DEVICE, TP_ERR_HIP, TP_ERR_CAPACITY, TP_ERR_STATE = range(6)
TRIANGULATE, WARP = 0, 1
IMAGE_A, IMAGE_B = 0, 1
BUF_TENERGY, BUF_COLNUM, BUF_COLACC, BUF_POINTS, BUF_GRADIENT, BUF_PENERGY, BUF_MOMENTS = range(7)

# every symbol include/tpose_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "tp_abi_version", "tp_device_count", "tp_last_error", "tp_create", "tp_destroy", "tp_set_ratio",
    "tp_get_ratio", "tp_set_dp", "tp_set_option", "tp_set_image", "tp_set_image_device", "tp_upload", "tp_accumulate",
    "tp_energy", "tp_shift", "tp_default_params", "tp_iterate", "tp_retrieve", "tp_retrieve_many", "tp_synchronize",
    "tp_get_stream", "tp_profile_iterate", "tp_profile_accumulate", "tp_get_info", "tp_selftest_walker", "tp_render",
    "tp_prepare", "tp_selftest_line", "tp_timer_start", "tp_timer_stop", "tp_iterate_until", "tp_band_mailbox_bytes",
    "tp_band_attach", "tp_band_mailbox_alloc", "tp_band_mailbox_free", "tp_band_mailbox_export", "tp_band_mailbox_import",
    "tp_band_mailbox_close", "tp_evaluate_triangles", "tp_selftest_variant", "tp_iterate_frames",
]


def band_mailbox_bytes(points, triangles):
    lib = load()
    lib.tp_band_mailbox_bytes.restype = C.c_size_t
    lib.tp_band_mailbox_bytes.argtypes = [C.c_int, C.c_int]
    return int(lib.tp_band_mailbox_bytes(points, triangles))


RENDER_AVERAGE, RENDER_STORED = 0, 1
OPT_PERSISTENT, OPT_INJECT_GIVE_UP = 1, 3
PERSIST_OFF, PERSIST_AUTO = 0, 1
FRAME_GO_ON, FRAME_STOP_REPLAY, FRAME_STOP = 0, 1, 2
INFO_PATCHES, INFO_PATCH_LDS, INFO_PATCH_LINES, INFO_PERSIST_LAUNCHES, INFO_PERSIST_ITERS, INFO_CENSUS, INFO_REPLANS = 2, 3, 4, 5, 6, 7, 8
INFO_PERSIST_FAILURES, INFO_BOX_FINEGRAINED, INFO_WARM_LAUNCHES = 9, 10, 11
INFO_RETRY_MS, INFO_PLAN_ROWS = 12, 13
INFO_REPLANS_BALANCE, INFO_PLAN_BALANCE_X1000, INFO_HEAVIEST_VERTEX_X1000 = 14, 15, 16


class Params(C.Structure):
    _fields_ = [("flavour", C.c_int32), ("image_slot", C.c_int32), ("rate", C.c_float), ("dp", C.c_float)]


class TposeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("tpose_hip error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load():
    """Load libtpose_hip.so; raises if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s is missing: build it with `python -m tpose_amd.build` "
                               "(there is no CPU fallback)" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        lib.tp_last_error.restype = C.c_char_p
        lib.tp_last_error.argtypes = [C.c_void_p]
        lib.tp_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        lib.tp_destroy.argtypes = [C.c_void_p]
        lib.tp_set_ratio.argtypes = [C.c_void_p, C.c_float]
        lib.tp_get_ratio.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        lib.tp_set_dp.argtypes = [C.c_void_p, C.c_float]
        lib.tp_set_option.argtypes = [C.c_void_p, C.c_int, C.c_int64]
        lib.tp_set_image.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        lib.tp_evaluate_triangles.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.tp_set_image_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        lib.tp_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        lib.tp_accumulate.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.tp_energy.argtypes = [C.c_void_p, C.c_int]
        lib.tp_shift.argtypes = [C.c_void_p, C.c_float]
        lib.tp_default_params.argtypes = [C.c_int, C.POINTER(Params)]
        lib.tp_default_params.restype = None
        lib.tp_iterate.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int]
        lib.tp_retrieve.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        lib.tp_retrieve_many.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.tp_synchronize.argtypes = [C.c_void_p]
        lib.tp_get_stream.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        lib.tp_profile_iterate.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int, C.POINTER(C.c_double)]
        lib.tp_profile_accumulate.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int, C.POINTER(C.c_double)]
        lib.tp_get_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]
        lib.tp_iterate_until.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int, C.c_double, C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_float)]
        lib.tp_timer_start.argtypes = [C.c_void_p]
        lib.tp_timer_stop.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        lib.tp_device_count.argtypes = [C.POINTER(C.c_int)]
        lib.tp_selftest_walker.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        lib.tp_render.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
        lib.tp_prepare.argtypes = [C.c_void_p, C.POINTER(Params)]
        lib.tp_selftest_line.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        lib.tp_selftest_variant.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        _lib = lib
    return _lib


def device_count():
    n = C.c_int(0)
    load().tp_device_count(C.byref(n))
    return n.value


def default_params(flavour, dp=0.0, rate=None, image_slot=None):
    p = Params()
    load().tp_default_params(flavour, C.byref(p))
    p.dp = dp
    if rate is not None:
        p.rate = rate
    if image_slot is not None:
        p.image_slot = image_slot
    return p


class Context:
    """One context per GPU (tpose::init .. tpose::quit)."""

    def __init__(self, device, width, height):
        self.lib = load()
        self.h = C.c_void_p()
        rc = self.lib.tp_create(device, width, height, C.byref(self.h))
        if rc != TP_OK:
            raise TposeError(rc, self.lib.tp_last_error(None).decode())
        self.W, self.H = width, height
        self.NT = self.NP = 0

    def _ck(self, rc):
        if rc != TP_OK:
            raise TposeError(rc, self.lib.tp_last_error(self.h).decode())

    def close(self):
        if self.h:
            self.lib.tp_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_ratio(self, ratio):
        self._ck(self.lib.tp_set_ratio(self.h, ratio))

    def set_dp(self, dp):
        self._ck(self.lib.tp_set_dp(self.h, dp))

    def set_option(self, option, value):
        self._ck(self.lib.tp_set_option(self.h, option, value))

    def set_persistent(self, on):
        self.set_option(OPT_PERSISTENT, PERSIST_AUTO if on else PERSIST_OFF)

    def band_attach(self, band, n_bands, mailboxes, bytes_each, points, triangles, patches_per_band=0):
        """mailboxes: device addresses (ints), one per band, as this process addresses them, each of band_mailbox_bytes(points,
        triangles) bytes or more (tp_band_attach)"""
        arr = (C.c_void_p * max(1, len(mailboxes)))(*[C.c_void_p(int(m)) for m in mailboxes])
        self.lib.tp_band_attach.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_size_t, C.c_int, C.c_int, C.c_int]
        self._ck(self.lib.tp_band_attach(self.h, band, n_bands, arr, C.c_size_t(bytes_each), points, triangles, patches_per_band))

    def band_mailbox_alloc(self, nbytes):
        """a zeroed mailbox in fine-grained memory of this context's device (tp_band_mailbox_alloc); returns its device address"""
        out = C.c_void_p()
        self.lib.tp_band_mailbox_alloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        self._ck(self.lib.tp_band_mailbox_alloc(self.h, C.c_size_t(nbytes), C.byref(out)))
        return out.value

    def band_mailbox_export(self, box):
        """the 64-byte handle another process imports (tp_band_mailbox_export)"""
        h = (C.c_ubyte * 64)()
        self.lib.tp_band_mailbox_export.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        self._ck(self.lib.tp_band_mailbox_export(self.h, C.c_void_p(box), h))
        return bytes(h)

    def band_mailbox_import(self, handle):
        out = C.c_void_p()
        buf = (C.c_ubyte * 64)(*handle)
        self.lib.tp_band_mailbox_import.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
        self._ck(self.lib.tp_band_mailbox_import(self.h, buf, C.byref(out)))
        return out.value

    def band_mailbox_close(self, box):
        self.lib.tp_band_mailbox_close.argtypes = [C.c_void_p, C.c_void_p]
        self._ck(self.lib.tp_band_mailbox_close(self.h, C.c_void_p(box)))

    def band_mailbox_free(self, box):
        self.lib.tp_band_mailbox_free.argtypes = [C.c_void_p, C.c_void_p]
        self._ck(self.lib.tp_band_mailbox_free(self.h, C.c_void_p(box)))

    def set_image(self, slot, img):
        img = np.ascontiguousarray(img, np.uint8)
        assert img.shape == (self.H, self.W, 4)
        self._ck(self.lib.tp_set_image(self.h, slot, img.ctypes.data, img.strides[0]))

    def set_image_device(self, slot, ptr, stride):
        self._ck(self.lib.tp_set_image_device(self.h, slot, C.c_void_p(ptr), stride))

    def upload(self, points, tris, colors=None):
        points = np.ascontiguousarray(points, np.float32)
        tris = np.ascontiguousarray(tris, np.int32)
        assert points.ndim == 2 and points.shape[1] == 2 and tris.ndim == 2 and tris.shape[1] == 4
        cptr = None
        if colors is not None:
            colors = np.ascontiguousarray(colors, np.int32)
            assert colors.shape == tris.shape
            cptr = colors.ctypes.data
        self._ck(self.lib.tp_upload(self.h, points.ctypes.data, points.shape[0], tris.ctypes.data,
                                    tris.shape[0], cptr))
        self.NP, self.NT = points.shape[0], tris.shape[0]

    def accumulate(self, flavour=TRIANGULATE, slot=IMAGE_A):
        self._ck(self.lib.tp_accumulate(self.h, flavour, slot))

    def energy(self, flavour=TRIANGULATE):
        self._ck(self.lib.tp_energy(self.h, flavour))

    def shift(self, rate):
        self._ck(self.lib.tp_shift(self.h, rate))

    def iterate(self, params, n):
        self._ck(self.lib.tp_iterate(self.h, C.byref(params), n))

    def prepare(self, params):
        """build the launch graph of the fused iteration now (tp_prepare)"""
        self._ck(self.lib.tp_prepare(self.h, C.byref(params)))

    def selftest_line(self, ends, H, rows):
        """ends int32[n,4] (Xa,Ya,Xb,Yb in 1/256 px), H int32[n] -> int32[n, rows+2]: ra, rb, crossing columns"""
        ends = np.ascontiguousarray(ends, np.int32)
        H = np.ascontiguousarray(H, np.int32)
        out = np.zeros((ends.shape[0], rows + 2), np.int32)
        self._ck(self.lib.tp_selftest_line(self.h, ends.ctypes.data_as(C.c_void_p), H.ctypes.data_as(C.c_void_p),
                                           ends.shape[0], rows, out.ctypes.data_as(C.c_void_p)))
        return out

    def selftest_variant(self, sums, meta):
        """sums uint64[n,3,4] line sums, meta int32[n,8] {dir x3, flips, flavour, r, g, b} -> int32[n,10] (tp_selftest_variant)"""
        sums = np.ascontiguousarray(sums, np.uint64)
        meta = np.ascontiguousarray(meta, np.int32)
        out = np.zeros((meta.shape[0], 10), np.int32)
        self._ck(self.lib.tp_selftest_variant(self.h, sums.ctypes.data_as(C.c_void_p), meta.ctypes.data_as(C.c_void_p), meta.shape[0],
                                              out.ctypes.data_as(C.c_void_p)))
        return out

    def profile_iterate(self, params, n):
        us = C.c_double(0)
        self._ck(self.lib.tp_profile_iterate(self.h, C.byref(params), n, C.byref(us)))
        return us.value

    def profile_accumulate(self, params, launches=64):
        us = C.c_double(0)
        self._ck(self.lib.tp_profile_accumulate(self.h, C.byref(params), launches, C.byref(us)))
        return us.value

    def synchronize(self):
        self._ck(self.lib.tp_synchronize(self.h))

    def render(self, source=0, points=None):
        """flat-shaded RGBA8 picture [H, W, 4] (tp_render); source 0: average colours of the last sweep,
        1: the uploaded colours; `points` overrides the vertex positions for this picture"""
        out = np.zeros((self.H, self.W, 4), np.uint8)
        pp = None
        if points is not None:
            pp = np.ascontiguousarray(points, np.float32)
            assert pp.shape == (self.NP, 2)
        self._ck(self.lib.tp_render(self.h, source, pp.ctypes.data if pp is not None else None,
                                    out.ctypes.data, out.strides[0]))
        return out

    def retrieve(self, what, count=None):
        V = 13 * self.NT
        shape, dtype = {
            BUF_TENERGY: ((V,), np.int32), BUF_COLNUM: ((V,), np.int32), BUF_COLACC: ((V, 4), np.int32),
            BUF_POINTS: ((self.NP, 2), np.float32), BUF_GRADIENT: ((self.NP, 2), np.int32),
            BUF_PENERGY: ((V,), np.int32), BUF_MOMENTS: ((V, 6), np.int64),
        }[what]
        out = np.zeros(shape, dtype)
        n = out.size if count is None else count
        self._ck(self.lib.tp_retrieve(self.h, what, out.ctypes.data, n))
        return out

    def evaluate_triangles(self, vertices, slot=IMAGE_A, variants=None):
        """base energy and pixel count each of the triangles `vertices` (n x 3 indices into the uploaded points) would have at the
        current positions (tp_evaluate_triangles); returns (energy, count) int32 arrays"""
        v = np.ascontiguousarray(vertices, np.int32).reshape(-1, 3)
        n = v.shape[0]
        e, cnt = np.zeros(n, np.int32), np.zeros(n, np.int32)
        va = None if variants is None else np.ascontiguousarray(variants, np.int32)
        assert va is None or va.shape == (n,)
        self._ck(self.lib.tp_evaluate_triangles(self.h, slot, n, v.ctypes.data, None if va is None else va.ctypes.data, e.ctypes.data, cnt.ctypes.data))
        return e, cnt

    def retrieve_many(self, whats):
        """several buffers with one wait (tp_retrieve_many); returns a list of arrays"""
        V = 13 * self.NT
        spec = {BUF_TENERGY: ((V,), np.int32), BUF_COLNUM: ((V,), np.int32), BUF_COLACC: ((V, 4), np.int32),
                BUF_POINTS: ((self.NP, 2), np.float32), BUF_GRADIENT: ((self.NP, 2), np.int32),
                BUF_PENERGY: ((V,), np.int32), BUF_MOMENTS: ((V, 6), np.int64)}
        outs = [np.zeros(*spec[w]) for w in whats]
        n = len(whats)
        wa = (C.c_int * n)(*whats)
        da = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        ca = (C.c_size_t * n)(*[o.size for o in outs])
        self._ck(self.lib.tp_retrieve_many(self.h, n, wa, da, ca))
        return outs

    def info(self, what):
        v = C.c_int64(0)
        self._ck(self.lib.tp_get_info(self.h, what, C.byref(v)))
        return v.value

    def selftest_walker(self, N0, step, d):
        N0 = np.ascontiguousarray(N0, np.int64)
        step = np.ascontiguousarray(step, np.int32)
        d = np.ascontiguousarray(d, np.int32)
        out = np.zeros((N0.shape[0], 32), np.int32)
        self._ck(self.lib.tp_selftest_walker(self.h, N0.ctypes.data, step.ctypes.data, d.ctypes.data,
                                             N0.shape[0], out.ctypes.data))
        return out

    def iterate_until(self, params, max_frames, threshold, toterr=1.0):
        """frames until tpose::geterr < threshold (or max_frames); returns (frames run, toterr, relerr of the last frame)"""
        tot, n, rel = C.c_float(toterr), C.c_int(0), C.c_float(0.0)
        self._ck(self.lib.tp_iterate_until(self.h, C.byref(params), max_frames, C.c_double(threshold), C.byref(tot), C.byref(n), C.byref(rel)))
        return n.value, tot.value, rel.value

    def iterate_frames(self, params, max_frames, fn):
        """tp_iterate_frames: fn(k, tenergy int32[NT], points float32[NP, 2]) -> FRAME_GO_ON / FRAME_STOP_REPLAY / FRAME_STOP after every frame
        (the arrays are views of the library's buffers: copy what is kept); returns the frames handed over"""
        NT, NP = self.NT, self.NP
        FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_float))

        def tramp(_user, k, ten, pts):
            return int(fn(k, np.ctypeslib.as_array(ten, shape=(NT,)), np.ctypeslib.as_array(pts, shape=(NP, 2))))
        cb = FN(tramp)
        n = C.c_int(0)
        self.lib.tp_iterate_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_int, FN, C.c_void_p, C.POINTER(C.c_int)]
        self._ck(self.lib.tp_iterate_frames(self.h, C.byref(params), max_frames, cb, None, C.byref(n)))
        return n.value

    def timer_start(self):
        self._ck(self.lib.tp_timer_start(self.h))

    def timer_stop(self):
        """microseconds between timer_start and now on the library's stream (HIP events); waits for the stream"""
        us = C.c_double(0.0)
        self._ck(self.lib.tp_timer_stop(self.h, C.byref(us)))
        return us.value

    def stream(self):
        s = C.c_void_p()
        self._ck(self.lib.tp_get_stream(self.h, C.byref(s)))
        return s.value
