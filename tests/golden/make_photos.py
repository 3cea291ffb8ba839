#!/usr/bin/env python
"""Decode the pictures BASELINE.json's configs name -- /root/reference/resource/{fruit,meninas,imageA,imageB,shoeA,shoeB}.png --
into raw fixtures under tests/golden/photos/ (DATA only: the decoded RGB bytes of the reference's input pictures; alpha is 255 everywhere
in all six and the shaders read .rgb only, software/triangulate/shader/triangle.fs:30,40).

Runs in the build container only (it needs /root/reference and Pillow); the fixtures travel, the PNGs do not.

Format of `<name>.rgb.xz`: LZMA (xz container) of the H x W x 3 u8 array after a horizontal difference modulo 256
(d[y, 0] = p[y, 0], d[y, x] = p[y, x] - p[y, x - 1]): tpose_amd/photos.py undoes it with a cumulative sum.  `index.json` holds
{name: {"w", "h", "sha256" of the raw RGB bytes, "source"}}.

  python tests/golden/make_photos.py
"""
import hashlib
import json
import lzma
import os

import numpy as np
from PIL import Image

SRC = "/root/reference/resource"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "photos")
NAMES = ["fruit", "meninas", "imageA", "imageB", "shoeA", "shoeB"]


def main():
    os.makedirs(OUT, exist_ok=True)
    index = {}
    for n in NAMES:
        a = np.asarray(Image.open(os.path.join(SRC, n + ".png")).convert("RGBA"))
        assert int(a[..., 3].min()) == 255, n
        rgb = np.ascontiguousarray(a[..., :3])
        d = rgb.copy()
        d[:, 1:] = rgb[:, 1:] - rgb[:, :-1]
        blob = lzma.compress(d.tobytes(), format=lzma.FORMAT_XZ, preset=9)
        with open(os.path.join(OUT, n + ".rgb.xz"), "wb") as f:
            f.write(blob)
        index[n] = {"w": int(rgb.shape[1]), "h": int(rgb.shape[0]), "sha256": hashlib.sha256(rgb.tobytes()).hexdigest(),
                    "source": "resource/%s.png" % n}
        print(n, rgb.shape, len(blob))
    with open(os.path.join(OUT, "index.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
