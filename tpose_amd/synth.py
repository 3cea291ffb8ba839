"""Deterministic synthetic inputs: RGBA8 rasters and jittered-grid
triangulations in the reference's data model (source/triangulation.hpp:26-69).

No network, no datasets: bench.py, smoke() and the tests all draw their inputs from here, so the
GPU path, the oracle and the CPU baseline see byte-identical data.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed, n):
    """n successive splitmix64 outputs (uint64) for `seed`; vectorised (counter form)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _uniform(seed, n):
    """float64 in [0,1)"""
    return (splitmix64(seed, n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def voronoi_raster(W, H, seed=1234, sites=96, noise=8):
    """RGBA8 (H, W, 4), A = 255: `sites`-cell Voronoi partition with uniform random u8 site colours
    plus per-pixel uniform noise in [-noise, noise] per channel, clamped."""
    u = _uniform(seed, sites * 5)
    sx = u[0:sites] * W
    sy = u[sites:2 * sites] * H
    col = np.floor(u[2 * sites:5 * sites].reshape(3, sites) * 256.0).astype(np.int32)
    img = np.empty((H, W, 4), np.uint8)
    xs = np.arange(W, dtype=np.float32) + 0.5
    band = max(1, (1 << 22) // max(W, 1))
    for r0 in range(0, H, band):
        r1 = min(H, r0 + band)
        ys = np.arange(r0, r1, dtype=np.float32) + 0.5
        best = np.full((r1 - r0, W), np.inf, np.float32)
        arg = np.zeros((r1 - r0, W), np.int32)
        for s in range(sites):
            d = (xs[None, :] - np.float32(sx[s])) ** 2 + (ys[:, None] - np.float32(sy[s])) ** 2
            m = d < best
            best[m] = d[m]
            arg[m] = s
        nz = splitmix64(seed ^ 0x5EED0000 ^ (r0 * 0x10001), (r1 - r0) * W * 3)
        nz = (nz % np.uint64(2 * noise + 1)).astype(np.int32).reshape(r1 - r0, W, 3) - noise
        for ch in range(3):
            img[r0:r1, :, ch] = np.clip(col[ch][arg] + nz[:, :, ch], 0, 255).astype(np.uint8)
    img[:, :, 3] = 255
    return img


def displaced_raster(img, amp=24.0):
    """Image B for the warp flavour: `img` resampled (nearest texel) under a fixed smooth
    displacement of at most `amp` pixels: dx = amp*sin(2*pi*y/H)*0.5, dy = amp*sin(2*pi*x/W)*0.5."""
    H, W = img.shape[:2]
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    dx = 0.5 * amp * np.sin(2 * np.pi * yy / H)
    dy = 0.5 * amp * np.sin(2 * np.pi * xx / W)
    sx = np.clip(np.rint(xx + dx), 0, W - 1).astype(np.int64)
    sy = np.clip(np.rint(yy + dy), 0, H - 1).astype(np.int64)
    return np.ascontiguousarray(img[sy, sx])


def build_halfedges(tris):
    """halfedges[3t+k] = id of the opposite half-edge or -1; half-edge k of triangle t runs from
    vertex k to vertex (k+1)%3 (source/triangulation.hpp:105-116, ctor :56-62)."""
    NT = tris.shape[0]
    he = np.full(3 * NT, -1, np.int32)
    edge = {}
    for t in range(NT):
        for k in range(3):
            edge[(int(tris[t, k]), int(tris[t, (k + 1) % 3]))] = 3 * t + k
    for (a, b), h in edge.items():
        he[h] = edge.get((b, a), -1)
    return he


def two_triangle(ratio):
    """The reference's 2-triangle start state (source/triangulation.hpp:42-69)."""
    pts = np.array([[-ratio, -1], [-ratio, 1], [ratio, -1], [ratio, 1]], np.float32)
    tris = np.array([[0, 1, 2, 0], [2, 1, 3, 0]], np.int32)
    he = np.array([-1, 3, -1, 1, -1, -1], np.int32)
    return pts, tris, he


def grid_triangulation(gx, gy, ratio=1.0, jitter=0.25, seed=99):
    """Jittered-grid triangulation with gx*gy*2 triangles.

    Points 0..3 are the corners in reference order (-R,-1), (-R,1), (R,-1), (R,1); triangles are
    clockwise in the y-up t-pose space like the reference's (0,1,2), (2,1,3); interior vertices
    are jittered by +-`jitter` cells; boundary vertices stay on the boundary.
    Returns points f32[NP,2], triangles i32[NT,4] (w = 0), halfedges i32[3*NT].
    """
    nx, ny = gx + 1, gy + 1
    u = _uniform(seed, nx * ny * 2).reshape(2, ny, nx)
    X = np.empty((ny, nx), np.float64)
    Y = np.empty((ny, nx), np.float64)
    cw, ch = 2.0 * ratio / gx, 2.0 / gy
    for j in range(ny):
        for i in range(nx):
            x, y = -ratio + i * cw, -1.0 + j * ch
            if 0 < i < gx and 0 < j < gy:
                x += (u[0, j, i] * 2 - 1) * jitter * cw
                y += (u[1, j, i] * 2 - 1) * jitter * ch
            X[j, i], Y[j, i] = x, y
    vid = -np.ones((ny, nx), np.int64)
    vid[0, 0], vid[gy, 0], vid[0, gx], vid[gy, gx] = 0, 1, 2, 3
    nxt = 4
    for j in range(ny):
        for i in range(nx):
            if vid[j, i] < 0:
                vid[j, i] = nxt
                nxt += 1
    pts = np.zeros((nx * ny, 2), np.float32)
    pts[vid.ravel(), 0] = X.ravel().astype(np.float32)
    pts[vid.ravel(), 1] = Y.ravel().astype(np.float32)
    tris = np.zeros((gx * gy * 2, 4), np.int32)
    k = 0
    for j in range(gy):
        for i in range(gx):
            a, b, c, d = vid[j, i], vid[j + 1, i], vid[j, i + 1], vid[j + 1, i + 1]
            tris[k, :3] = (a, b, c)      # like (0,1,2): (-,-) -> (-,+) -> (+,-), clockwise y-up
            tris[k + 1, :3] = (c, b, d)  # like (2,1,3)
            k += 2
    return pts, tris, build_halfedges(tris)


def mean_colors(img, pts, tris, ratio):
    """Per-triangle stored colours (ivec4, w = 1) for the warp flavour: colour of the texel under
    each triangle's centroid (cheap stand-in for a converged .tri; any 0..255 colours do)."""
    H, W = img.shape[:2]
    c = pts[tris[:, :3]].mean(axis=1)
    col = np.clip(((c[:, 0] / ratio + 1) * 0.5 * W).astype(np.int64), 0, W - 1)
    row = np.clip(((1 - c[:, 1]) * 0.5 * H).astype(np.int64), 0, H - 1)
    out = np.ones((tris.shape[0], 4), np.int32)
    out[:, :3] = img[row, col, :3]
    return out


GRID_FOR_NT = {3000: (50, 30), 12000: (100, 60), 150: (15, 5), 48: (6, 4), 2: None}


def photo_contrast(img, contrast):
    """the raster with its RGB contrast about mid-grey scaled by `contrast` (rounded to nearest)"""
    out = img.copy()
    rgb = img[:, :, :3].astype(np.float32)
    out[:, :, :3] = np.clip(128.0 + (rgb - 128.0) * np.float32(contrast) + 0.5, 0, 255).astype(np.uint8)
    return out


def workload(W, H, NT, seed=1234, contrast=1.0):
    """(imgA, points, triangles, halfedges, ratio) for a named synthetic workload.  contrast < 1: the same Voronoi +
    noise raster with photograph-like contrast between neighbouring regions (the reference's fixed-step descent,
    shift.cs:45, is only stable on such rasters: a full-contrast one throws vertices hundreds of pixels)."""
    ratio = float(np.float32(W) / np.float32(H))
    img = voronoi_raster(W, H, seed=seed)
    if contrast != 1.0:
        img = photo_contrast(img, contrast)
    if NT == 2:
        pts, tris, he = two_triangle(ratio)
    else:
        gx, gy = GRID_FOR_NT[NT]
        pts, tris, he = grid_triangulation(gx, gy, ratio=ratio)
    return img, pts, tris, he, ratio
