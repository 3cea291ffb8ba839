"""ctypes binding of libtpose_host.so: the C++ host mirror (tpose::triangulation, tpose::io) for the
Python multi-GPU drivers.  Host-only; no GPU, no HIP."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "host", "libtpose_host.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        # make decides staleness (host_capi.cpp AND the include/tpose/*.hpp headers are its prerequisites); ranks started
        # together (torchrun) must not build into the same file at once: one builds under an exclusive lock, the rest wait
        import fcntl
        try:
            with open(os.path.join(_HERE, "host", ".build.lock"), "w") as lock:
                fcntl.flock(lock, fcntl.LOCK_EX)
                try:
                    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "host"), "libtpose_host.so"])
                finally:
                    fcntl.flock(lock, fcntl.LOCK_UN)
        except (OSError, subprocess.CalledProcessError):
            # a read-only install, or a box without make: the library that is there is the one to load
            if not os.path.exists(_SO):
                raise
        lib = C.CDLL(_SO)
        lib.tph_new.restype = C.c_void_p
        lib.tph_get_ratio.restype = C.c_float
        lib.tph_set_ratio.argtypes = [C.c_float]
        for f in ("tph_triangles", "tph_halfedges", "tph_colors"):
            getattr(lib, f).restype = C.POINTER(C.c_int32)
            getattr(lib, f).argtypes = [C.c_void_p]
        for f in ("tph_points", "tph_originpoints"):
            getattr(lib, f).restype = C.POINTER(C.c_float)
            getattr(lib, f).argtypes = [C.c_void_p]
        for f in ("tph_free", "tph_nt", "tph_np", "tph_points_from_origin", "tph_origin_from_points"):
            getattr(lib, f).argtypes = [C.c_void_p]
        lib.tph_assign.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5
        lib.tph_set_points.argtypes = [C.c_void_p, C.c_void_p]
        lib.tph_read.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        lib.tph_write.argtypes = [C.c_void_p, C.c_char_p]
        lib.tph_warp.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        lib.tph_reversewarp.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        lib.tph_flip.argtypes = [C.c_void_p, C.c_int, C.c_float]
        for f in ("tph_split", "tph_collapse", "tph_prune"):
            getattr(lib, f).argtypes = [C.c_void_p, C.c_int]
        lib.tph_fundamental.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        lib.tph_mean_sampson.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        lib.tph_mean_sampson.restype = C.c_double
        lib.tph_correct_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        lib.tph_triangulate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        lib.tph_realroots.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.tph_epole.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.tph_readmatches.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_int]
        lib.tph_set_verbose(0)
        _lib = lib
    return _lib


def set_ratio(r):
    load().tph_set_ratio(r)


def get_ratio():
    return float(load().tph_get_ratio())


class Triangulation:
    """tpose::triangulation (include/tpose/triangulation.hpp)."""

    def __init__(self):
        self.lib = load()
        self.h = C.c_void_p(self.lib.tph_new())

    def __del__(self):
        try:
            if self.h:
                self.lib.tph_free(self.h)
                self.h = None
        except Exception:
            pass

    @property
    def NT(self):
        return self.lib.tph_nt(self.h)

    @property
    def NP(self):
        return self.lib.tph_np(self.h)

    def _arr(self, fn, n, dtype):
        ptr = getattr(self.lib, fn)(self.h)
        return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)

    @property
    def triangles(self):
        return self._arr("tph_triangles", 4 * self.NT, np.int32).reshape(-1, 4)

    @property
    def halfedges(self):
        return self._arr("tph_halfedges", 3 * self.NT, np.int32)

    @property
    def colors(self):
        return self._arr("tph_colors", 4 * self.NT, np.int32).reshape(-1, 4)

    @property
    def points(self):
        return self._arr("tph_points", 2 * self.NP, np.float32).reshape(-1, 2)

    @points.setter
    def points(self, pts):
        pts = np.ascontiguousarray(pts, np.float32)
        assert pts.shape == (self.NP, 2)
        self.lib.tph_set_points(self.h, pts.ctypes.data)

    @property
    def originpoints(self):
        return self._arr("tph_originpoints", 2 * self.NP, np.float32).reshape(-1, 2)

    def assign(self, tris, points, originpoints, halfedges=None, colors=None):
        tris = np.ascontiguousarray(tris, np.int32)
        points = np.ascontiguousarray(points, np.float32)
        originpoints = np.ascontiguousarray(originpoints, np.float32)
        he = None if halfedges is None else np.ascontiguousarray(halfedges, np.int32)
        co = None if colors is None else np.ascontiguousarray(colors, np.int32)
        self.lib.tph_assign(self.h, tris.shape[0], points.shape[0], tris.ctypes.data, points.ctypes.data,
                            originpoints.ctypes.data, None if he is None else he.ctypes.data,
                            None if co is None else co.ctypes.data)

    def points_from_origin(self):
        self.lib.tph_points_from_origin(self.h)

    def origin_from_points(self):
        self.lib.tph_origin_from_points(self.h)

    def read(self, path, dowarp=False):
        return bool(self.lib.tph_read(self.h, path.encode(), int(dowarp)))

    def write(self, path):
        self.lib.tph_write(self.h, path.encode())

    def warp(self, pts):
        pts = np.array(pts, np.float32, copy=True)
        self.lib.tph_warp(self.h, pts.ctypes.data, pts.shape[0])
        return pts

    def reversewarp(self, pts):
        pts = np.array(pts, np.float32, copy=True)
        self.lib.tph_reversewarp(self.h, pts.ctypes.data, pts.shape[0])
        return pts

    def flip(self, h, minangle=3.14159265):
        return bool(self.lib.tph_flip(self.h, h, minangle))

    def split(self, t):
        return bool(self.lib.tph_split(self.h, t))

    def collapse(self, h):
        return bool(self.lib.tph_collapse(self.h, h))


# ---- two-view geometry (include/tpose/multiview.hpp) ------------------------------------------------
F_8POINT, F_SAMPSON, F_RANSAC, F_LMEDS = 0, 1, 2, 3


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


def fundamental(method, A, B):
    """tpose::mview::F_8Point / F_Sampson / F_RANSAC / F_LMEDS on matches A, B: float[N, 2] -> F[3, 3]"""
    A, B = _f32(A), _f32(B)
    F = np.zeros((3, 3), np.float32)
    load().tph_fundamental(method, A.ctypes.data, B.ctypes.data, A.shape[0], F.ctypes.data)
    return F


def mean_sampson(F, A, B):
    F, A, B = _f32(F), _f32(A), _f32(B)
    return float(load().tph_mean_sampson(F.ctypes.data, A.ctypes.data, B.ctypes.data, A.shape[0]))


def correct_matches(F, A, B):
    """optimal two-view correction (tpose::mview::triangulate(F, A, B)); returns the moved copies"""
    F, A, B = _f32(F), _f32(A).copy(), _f32(B).copy()
    load().tph_correct_matches(F.ctypes.data, A.ctypes.data, B.ctypes.data, A.shape[0])
    return A, B


def triangulate(F, K, A, B, check=3):
    F, K, A, B = _f32(F), _f32(K), _f32(A), _f32(B)
    X = np.zeros((A.shape[0], 4), np.float32)
    load().tph_triangulate(F.ctypes.data, K.ctypes.data, A.ctypes.data, B.ctypes.data, A.shape[0], check, X.ctypes.data)
    return X


def realroots(coeff):
    c = np.ascontiguousarray(coeff, np.float64)
    out = np.zeros(c.size, np.float64)
    n = load().tph_realroots(c.ctypes.data, c.size, out.ctypes.data)
    return out[:n]


def epole(F, right=True):
    F = _f32(F)
    e = np.zeros(2, np.float32)
    load().tph_epole(F.ctypes.data, 1 if right else 0, e.ctypes.data)
    return e


def readmatches(path, cap=1 << 16):
    a, b = np.zeros((cap, 2), np.float32), np.zeros((cap, 2), np.float32)
    n = load().tph_readmatches(path.encode(), a.ctypes.data, b.ctypes.data, cap)
    if n < 0:
        raise IOError(path)
    return a[:n].copy(), b[:n].copy()
