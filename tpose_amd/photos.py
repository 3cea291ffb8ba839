"""The pictures BASELINE.json's configs name (resource/fruit.png, meninas.png, imageA/B.png, shoeA/B.png of the reference), as decoded
fixtures under tests/golden/photos/ (made by tests/golden/make_photos.py in the build container; data only), and an integer resampler.

Input data for tests and for bench.py's photograph figure -- nothing here computes anything of the hot path."""
import hashlib
import json
import lzma
import os

import numpy as np

DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "photos")
NAMES = ("fruit", "meninas", "imageA", "imageB", "shoeA", "shoeB")
_cache = {}


def index():
    with open(os.path.join(DIR, "index.json")) as f:
        return json.load(f)


def load(name, verify=True):
    """RGBA8 (H, W, 4), A = 255: the decoded picture `name` at its own size."""
    if name in _cache:
        return _cache[name].copy()
    meta = index()[name]
    with open(os.path.join(DIR, name + ".rgb.xz"), "rb") as f:
        raw = lzma.decompress(f.read(), format=lzma.FORMAT_XZ)
    d = np.frombuffer(raw, np.uint8).reshape(meta["h"], meta["w"], 3)
    rgb = np.cumsum(d, axis=1, dtype=np.uint8)          # (undo the horizontal difference, modulo 256)
    if verify and hashlib.sha256(rgb.tobytes()).hexdigest() != meta["sha256"]:
        raise ValueError("fixture %s does not decode to the bytes its index names" % name)
    img = np.empty((meta["h"], meta["w"], 4), np.uint8)
    img[:, :, :3] = rgb
    img[:, :, 3] = 255
    _cache[name] = img
    return img.copy()


def resample_int(img, W2, H2):
    """Bilinear resample in INTEGER arithmetic (texel centres at +0.5, clamp to edge, weights in 1/256, rounded to nearest):
    out[y, x] = (sum of the four neighbours x their 8.8 weights + 2^15) >> 16.  The same bytes on every machine."""
    H, W = img.shape[:2]

    def axis(n_src, n_dst):
        num = (2 * np.arange(n_dst, dtype=np.int64) + 1) * n_src - n_dst    # (u + 0.5) * 2 n_dst, u the source coordinate of a centre
        den = 2 * n_dst
        i0 = np.floor_divide(num, den)
        fr = ((num - i0 * den) * 256) // den                                   # 0 .. 255
        return np.clip(i0, 0, n_src - 1), np.clip(i0 + 1, 0, n_src - 1), fr

    x0, x1, fx = axis(W, W2)
    y0, y1, fy = axis(H, H2)
    src = img.astype(np.int64)
    out = np.empty((H2, W2, img.shape[2]), np.uint8)
    band = max(1, (1 << 20) // max(W2, 1))
    for r0 in range(0, H2, band):
        r1 = min(H2, r0 + band)
        a = src[y0[r0:r1]][:, x0]; b = src[y0[r0:r1]][:, x1]
        c = src[y1[r0:r1]][:, x0]; d = src[y1[r0:r1]][:, x1]
        wx = fx[None, :, None]; wy = fy[r0:r1, None, None]
        top = a * (256 - wx) + b * wx
        bot = c * (256 - wx) + d * wx
        out[r0:r1] = ((top * (256 - wy) + bot * wy + 32768) >> 16).astype(np.uint8)
    return out


def window(name, divisor_num=3, divisor_den=2):
    """The raster the reference rasterises into for picture `name`: its window of image / 1.5, truncated
    (software/triangulate/main.cpp:53) -- here by the integer resampler above."""
    img = load(name)
    H, W = img.shape[:2]
    return resample_int(img, W * divisor_den // divisor_num, H * divisor_den // divisor_num)


def raster_from_env(W, H, NT, default_contrast=0.1):
    """(img, points, triangles, halfedges, ratio, label) of a timing tool's workload: the synthetic raster at TPOSE_CONTRAST (default
    `default_contrast`), or -- TPOSE_PHOTO=name -- one of the reference's pictures resampled to W x H, under the same jittered grid."""
    from . import synth
    name = os.environ.get("TPOSE_PHOTO", "")
    contrast = float(os.environ.get("TPOSE_CONTRAST", str(default_contrast)))
    img, pts, tris, he, ratio = synth.workload(W, H, NT, contrast=contrast)
    if name:
        return resample_int(load(name), W, H), pts, tris, he, ratio, "photo %s resampled to %dx%d" % (name, W, H)
    return img, pts, tris, he, ratio, "synthetic, contrast %g" % contrast
