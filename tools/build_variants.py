#!/usr/bin/env python
"""Build A/B variants of libtpose_hip.so for timing experiments on the GPU box (tools/time_acc.py picks one with
TPOSE_HIP_LIB).  Variants live in tpose_amd/variants/ (git-ignored like every .so; they travel with gpurun).
  python tools/build_variants.py name=-DFLAG[,-DFLAG2] ...
The product library is never built with these flags."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tpose_amd import build as tb  # noqa: E402

out_dir = os.path.join(ROOT, "tpose_amd", "variants")
os.makedirs(out_dir, exist_ok=True)
for arg in sys.argv[1:]:
    name, _, flags = arg.partition("=")
    lib = os.path.join(out_dir, "libtpose_hip_%s.so" % name)
    tb.build(force=True, verbose=False, extra=[f for f in flags.split(",") if f], out=lib)
    print(lib)
