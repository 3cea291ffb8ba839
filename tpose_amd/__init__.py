"""tpose_amd -- MI355X-native (gfx950) implementation of the t-pose hot path: per-triangle
min-energy triangulation / warp gradient descent behind the reference's vertex-and-index API.

  csrc/            hand-written HIP kernels + the C ABI (include/tpose_hip.h)
  capi             ctypes plumbing over that ABI (tests, bench, multi-GPU drivers)
  synth            deterministic synthetic inputs
  build            in-tree hipcc build of libtpose_hip.so

The C++ host mirror of the reference interface (tpose::triangulation, tpose::io, tpose::upload ...)
lives in include/tpose/.
"""
from . import capi, synth  # noqa: F401

__all__ = ["capi", "synth"]
