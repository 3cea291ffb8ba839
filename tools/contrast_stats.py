#!/usr/bin/env python
"""Where does bench.py's raster contrast (x0.1 about mid-grey) sit among the reference's own photographs?  CONTAINER ONLY: reads
the pictures under /root/reference/resource (they do not travel; only the statistics below are committed, profiles/r04_contrast_stats.txt).

What drives the fixed-step descent (shift.cs:45: p -= rate * gradient / 65536) is the colour contrast between neighbouring regions at
the scale of a triangle: the gradient of a vertex is a difference of energies of variants displaced by dp, i.e. (colour step across an
edge)^2 x (pixels swept).  So, per picture at the window the reference opens (image / 1.5, software/triangulate/main.cpp:53) and at the
3000 triangles of the headline configuration (README.md:35), with blocks of the area of two triangles:
  within  = mean over blocks of the summed RGB variance inside a block      (what the energy per pixel measures: 2 E / pixels)
  between = mean squared difference of the RGB means of adjacent blocks     (what the gradients are made of)
  drive   = between x block^2: the step is taken in t-pose units whatever the raster's size (shift.cs:45 has no normalisation -- README.md:139
            calls it "very hard because of integer atomics"), so a vertex of a 2048^2 raster, whose variants sweep (2048 / 800)^2 times the
            pixels, moves that much further per grad-iter on the same contrast: THIS is what has to match for the descent to behave like the
            reference's on its photographs
The synthetic Voronoi + noise raster of SURVEY section 8(d) is measured the same way at contrasts 1.0 (rounds 1-2), 0.3, 0.14 and 0.1 (bench.py).
"""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tpose_amd import synth  # noqa: E402

REF = "/root/reference/resource"
NT = 3000


def stats(rgb):
    H, W = rgb.shape[:2]
    B = max(2, int(round((2.0 * W * H / NT) ** 0.5)))
    h, w = (H // B) * B, (W // B) * B
    x = rgb[:h, :w].astype(np.float64).reshape(h // B, B, w // B, B, 3)
    mean = x.mean(axis=(1, 3))
    within = ((x - mean[:, None, :, None, :]) ** 2).mean(axis=(1, 3)).sum(axis=2).mean()
    dx = ((mean[:, 1:] - mean[:, :-1]) ** 2).sum(axis=2).mean()
    dy = ((mean[1:] - mean[:-1]) ** 2).sum(axis=2).mean()
    return B, within, 0.5 * (dx + dy)


rows = []
for name in ("meninas.png", "fruit.png", "imageA.png", "imageB.png", "canyon.png", "shoeA.png", "shoeB.png"):
    p = os.path.join(REF, name)
    if not os.path.exists(p):
        continue
    im = Image.open(p).convert("RGB")
    w, h = int(im.width / 1.5), int(im.height / 1.5)
    rgb = np.asarray(im.resize((w, h), Image.BILINEAR))
    rows.append(("reference " + name, w, h) + stats(rgb))
for c in (1.0, 0.3, 0.14, 0.1):
    img = synth.workload(2048, 2048, NT, contrast=c)[0]
    rows.append(("synthetic Voronoi+noise, contrast %.2f" % c, 2048, 2048) + stats(img[:, :, :3]))
print("%-42s %6s %6s %5s %12s %12s %14s" % ("raster", "W", "H", "block", "within", "between", "drive"))
for r in rows:
    print("%-42s %6d %6d %5d %12.1f %12.1f %14.0f" % (r + (r[5] * r[3] ** 2,)))
ref_b = [r[5] for r in rows if r[0].startswith("reference")]
ref_d = [r[5] * r[3] ** 2 for r in rows if r[0].startswith("reference")]
print("\nthe reference's photographs: between-block contrast min %.1f  median %.1f  max %.1f; drive min %.0f  median %.0f  max %.0f"
      % (min(ref_b), float(np.median(ref_b)), max(ref_b), min(ref_d), float(np.median(ref_d)), max(ref_d)))
for r in rows:
    if r[0].startswith("synthetic"):
        print("%s: contrast %.2f x, drive %.2f x the photographs' median" % (r[0], r[5] / float(np.median(ref_b)), r[5] * r[3] ** 2 / float(np.median(ref_d))))
