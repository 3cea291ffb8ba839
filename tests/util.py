"""Shared helpers for the test-suite."""
import numpy as np

from oracle import oracle as O
from tpose_amd import synth

RATE = {0: 0.00005, 1: 0.00003}


def case(W, H, grid, seed=7, sites=12):
    """(imgA, imgB, points, tris, ratio, colors)"""
    img = synth.voronoi_raster(W, H, seed=seed, sites=sites)
    imgB = synth.displaced_raster(img, amp=6.0)
    ratio = float(np.float32(W) / np.float32(H))
    if grid is None:
        pts, tris, _ = synth.two_triangle(ratio)
    else:
        pts, tris, _ = synth.grid_triangulation(grid[0], grid[1], ratio=ratio)
    colors = synth.mean_colors(img, pts, tris, ratio)
    return img, imgB, pts, tris, ratio, colors


def oracle_step(img, pts, tris, flavour, ratio, colors=None, dp=None, literal=True):
    """One grad-iter through the oracle; returns dict like oracle.iterate."""
    return O.iterate(img, pts, tris, flavour, ratio, RATE[flavour], 1,
                     colors=colors if flavour == 1 else None, dp_=dp, literal=literal)
