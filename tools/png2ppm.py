#!/usr/bin/env python
"""PNG (or anything Pillow reads) -> binary PPM for the headless harnesses (tpose_amd/host)."""
import sys

from PIL import Image

if len(sys.argv) != 3:
    sys.exit("usage: png2ppm.py in.png out.ppm")
Image.open(sys.argv[1]).convert("RGB").save(sys.argv[2], format="PPM")
