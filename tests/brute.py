"""Independent NumPy brute-force evaluation of the hot path (tests only).

Shares no code with oracle/tp_oracle.c or the HIP kernels: vertex stage in numpy float32,
coverage by vectorised int64 edge functions over the whole raster, energies per fragment.
"""
import numpy as np

F = np.float32


def dp_law(flavour, NT):
    if flavour == 0:
        return F(0.05) / (F(1.0) + F(4.0) * F(NT) / F(3000.0))
    return F(0.05) / (F(1.0) + F(9.0) * F(NT) / F(1000.0))


def snap(f):
    v = F(f) * F(256.0) + F(0.5)
    v = np.fmin(np.fmax(v, F(-4194304.0)), F(8388608.0))
    return int(np.floor(v))


def variant_xy(points, tri, i, dp, ratio, W, H):
    out = []
    for s in range(3):
        px, py = F(points[tri[s], 0]), F(points[tri[s], 1])
        dx = dy = F(0)
        if i > 0 and (i - 1) // 4 == s:
            k = (i - 1) % 4
            dx, dy = [(F(dp), F(0)), (F(-dp), F(0)), (F(0), F(dp)), (F(0), F(-dp))][k]
        tx, ty = F(px + dx), F(py + dy)
        nx = F(tx / F(ratio))
        fx = F(F(nx + F(1)) * F(F(0.5) * F(W)))
        fy = F(F(F(1) - ty) * F(F(0.5) * F(H)))
        out += [snap(fx), snap(fy)]
    return out


def coverage_mask(xy, W, H):
    X0, Y0, X1, Y1, X2, Y2 = [int(v) for v in xy]
    area2 = (X1 - X0) * (Y2 - Y0) - (Y1 - Y0) * (X2 - X0)
    if area2 == 0:
        return np.zeros((H, W), bool)
    sg = 1 if area2 > 0 else -1
    px = (256 * np.arange(W, dtype=np.int64) + 128)[None, :]
    py = (256 * np.arange(H, dtype=np.int64) + 128)[:, None]
    m = np.ones((H, W), bool)
    V = [(X0, Y0), (X1, Y1), (X2, Y2)]
    for e in range(3):
        (xa, ya), (xb, yb) = V[e], V[(e + 1) % 3]
        a, b = -(yb - ya) * sg, (xb - xa) * sg
        E = a * (px - xa) + b * (py - ya)
        tl = a > 0 or (a == 0 and b > 0)
        m &= (E > 0) | ((E == 0) & tl)
    return m


def wrap32(x):
    return np.int64(x).astype(np.uint64).astype(np.uint32).astype(np.int32) if np.ndim(x) else \
        np.array(int(x) & 0xFFFFFFFF, np.uint32).astype(np.int32)[()]


def evaluate(img, points, tris, flavour, ratio, colors=None, dp=None):
    """cn, ca, ten (reference layout, variant-major id = i*NT + t), brute force."""
    H, W = img.shape[:2]
    NT = tris.shape[0]
    if dp is None:
        dp = dp_law(flavour, NT)
    rgb = img[:, :, :3].astype(np.int64)
    cn = np.zeros(13 * NT, np.int32)
    ca = np.zeros((13 * NT, 4), np.int32)
    ten = np.zeros(13 * NT, np.int32)
    for i in range(13):
        for t in range(NT):
            idx = i * NT + t
            m = coverage_mask(variant_xy(points, tris[t], i, dp, ratio, W, H), W, H)
            px = rgb[m]
            n = px.shape[0]
            cn[idx] = n
            if flavour == 0:
                s = px.sum(axis=0)
                ca[idx, :3] = s
                if n == 0:
                    continue
                a = s // n
            else:
                ca[idx] = colors[t]
                a = colors[t, :3].astype(np.int64)
            d2 = ((px - a[None, :]) ** 2).sum(axis=1)
            ten[idx] = wrap32((d2 >> 1).sum())
    return cn, ca, ten
