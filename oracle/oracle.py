"""ctypes binding of the CPU oracle (oracle/libtp_oracle.so).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED -- see oracle/tp_oracle.h.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product package (tpose_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

TRIANGULATE, WARP = 0, 1


class _Raster(C.Structure):
    _fields_ = [("rgba", C.c_void_p), ("stride", C.c_size_t), ("W", C.c_int), ("H", C.c_int)]


def build(force=False):
    so = os.path.join(_HERE, "libtp_oracle.so")
    src = [os.path.join(_HERE, f) for f in ("tp_oracle.c", "tp_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B" if force else "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.tpo_dp.restype = C.c_float
        _LIB.tpo_dp.argtypes = [C.c_int, C.c_int]
        _LIB.tpo_covered.restype = C.c_int
        for f in ("tpo_geterr", "tpo_gettoterr"):
            getattr(_LIB, f).restype = C.c_float
        _LIB.tpo_maxerrid.restype = C.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _raster(img):
    assert img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 4 and img.flags.c_contiguous
    return _Raster(img.ctypes.data, img.strides[0], img.shape[1], img.shape[0])


def dp(flavour, NT):
    return float(np.float32(lib().tpo_dp(flavour, NT)))


def variant_vertices(points, tris, t, i, dp_, ratio, W, H):
    out = np.zeros(6, np.int32)
    lib().tpo_variant_vertices(_p(points), _p(tris), C.c_int(t), C.c_int(i), C.c_float(dp_),
                               C.c_float(ratio), C.c_int(W), C.c_int(H), _p(out))
    return out


def covered(xy, c, r):
    xy = np.ascontiguousarray(xy, np.int32)
    return bool(lib().tpo_covered(_p(xy), C.c_int(c), C.c_int(r)))


def accumulate_literal(img, points, tris, dp_, ratio, count_only=False, nthreads=1, ca=None):
    NT = tris.shape[0]
    cn = np.zeros(13 * NT, np.int32)
    if ca is None:
        ca = np.zeros((13 * NT, 4), np.int32)
    r = _raster(img)
    lib().tpo_accumulate_literal(C.byref(r), _p(points), _p(tris), C.c_int(NT), C.c_float(dp_),
                                 C.c_float(ratio), C.c_int(int(count_only)), _p(cn), _p(ca),
                                 C.c_int(nthreads))
    return cn, ca


def energy_literal(img, points, tris, dp_, ratio, flavour, cn, ca, nthreads=1):
    NT = tris.shape[0]
    ten = np.zeros(13 * NT, np.int32)
    r = _raster(img)
    lib().tpo_energy_literal(C.byref(r), _p(points), _p(tris), C.c_int(NT), C.c_float(dp_),
                             C.c_float(ratio), C.c_int(flavour), _p(cn), _p(ca), _p(ten),
                             C.c_int(nthreads))
    return ten


def moments(img, points, tris, dp_, ratio):
    NT = tris.shape[0]
    mom = np.zeros((13 * NT, 6), np.int64)
    r = _raster(img)
    lib().tpo_moments(C.byref(r), _p(points), _p(tris), C.c_int(NT), C.c_float(dp_),
                      C.c_float(ratio), _p(mom))
    return mom


def finalize(mom, NT, flavour, colors=None):
    ten = np.zeros(13 * NT, np.int32)
    cn = np.zeros(13 * NT, np.int32)
    ca = np.zeros((13 * NT, 4), np.int32)
    ten64 = np.zeros(13 * NT, np.int64)
    lib().tpo_finalize(_p(mom), C.c_int(NT), C.c_int(flavour), _p(colors), _p(ten), _p(cn),
                       _p(ca) if flavour == TRIANGULATE else None, _p(ten64))
    return ten, cn, ca, ten64


def gradient(ten, tris, NP):
    NT = tris.shape[0]
    gr = np.zeros((NP, 2), np.int32)
    lib().tpo_gradient(_p(ten), _p(tris), C.c_int(NT), C.c_int(NP), _p(gr))
    return gr


def shift(points, gr, ratio, rate):
    pts = np.array(points, np.float32, copy=True)
    lib().tpo_shift(_p(pts), C.c_int(pts.shape[0]), _p(gr), C.c_float(ratio), C.c_float(rate))
    return pts


def iterate(img, points, tris, flavour, ratio, rate, iters, colors=None, dp_=None, literal=True,
            nthreads=1):
    """Run `iters` grad-iters; returns dict(points, ten, cn, ca, gr) after the last iteration."""
    NT, NP = tris.shape[0], points.shape[0]
    if dp_ is None:
        dp_ = dp(flavour, NT)
    pts = np.array(points, np.float32, copy=True)
    ten = np.zeros(13 * NT, np.int32)
    cn = np.zeros(13 * NT, np.int32)
    ca = np.zeros((13 * NT, 4), np.int32)
    gr = np.zeros((NP, 2), np.int32)
    r = _raster(img)
    if literal:
        lib().tpo_iterate_literal(C.byref(r), _p(pts), C.c_int(NP), _p(tris), C.c_int(NT),
                                  C.c_int(flavour), _p(colors), C.c_float(dp_), C.c_float(ratio),
                                  C.c_float(rate), C.c_int(iters), _p(ten), _p(cn), _p(ca), _p(gr),
                                  C.c_int(nthreads))
    else:
        lib().tpo_iterate_moments(C.byref(r), _p(pts), C.c_int(NP), _p(tris), C.c_int(NT),
                                  C.c_int(flavour), _p(colors), C.c_float(dp_), C.c_float(ratio),
                                  C.c_float(rate), C.c_int(iters), _p(ten), _p(cn), _p(ca), _p(gr))
    return dict(points=pts, ten=ten, cn=cn, ca=ca, gr=gr)


class ErrState:
    """tpose::toterr/newerr/relerr/maxerr globals (source/triangulation.hpp:648-651)."""

    def __init__(self):
        self.st = np.array([1.0, 0.0, 0.0, 0.0], np.float32)

    def geterr(self, terr, NT):
        return float(lib().tpo_geterr(_p(terr), C.c_int(NT), _p(self.st)))

    def gettoterr(self, terr, NT):
        return float(lib().tpo_gettoterr(_p(terr), C.c_int(NT), _p(self.st)))

    def maxerrid(self, terr, NT):
        return int(lib().tpo_maxerrid(_p(terr), C.c_int(NT), _p(self.st)))
