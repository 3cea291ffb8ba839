"""Build libtpose_hip.so (the C-ABI library with the gfx950 kernels) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the development container; the built
.so travels to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtpose_hip.so")
SOURCES = ["tp_kernels.hip", "tp_persist.hip", "tp_context.hip", "tp_persist_host.hip", "tp_replan.hip", "tp_bands.hip", "tp_readback.hip", "tp_eval.hip"]
HEADERS = ["tp_raster.h", "tp_kernels.h", "tp_plan.h", "tp_persist.h", "tp_context.h", os.path.join("..", "..", "include", "tpose_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built (no CPU fallback exists)")
    return exe


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra=(), out=None):
    """out: build a variant (debug flavour, timing experiments) next to the product library instead of it.
    The translation units are compiled side by side (one hipcc each, objects under tpose_amd/_obj/<flags>/, reused while
    neither their source nor any header is newer) and linked into one shared library."""
    if out is None and not force and not stale():
        return LIB
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    flags = [f for f in FLAGS if f != "-shared"] + list(extra)
    tag = hashlib.sha1(" ".join(flags).encode()).hexdigest()[:10]
    odir = os.path.join(HERE, "_obj", tag)
    os.makedirs(odir, exist_ok=True)
    hdr_time = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    hdr_time = max(hdr_time, os.path.getmtime(os.path.abspath(__file__)))

    def compile_one(src):
        obj = os.path.join(odir, os.path.splitext(src)[0] + ".o")
        path = os.path.join(CSRC, src)
        if os.path.exists(obj) and os.path.getmtime(obj) > max(hdr_time, os.path.getmtime(path)):
            return obj
        cmd = [hipcc()] + flags + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out or LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out or LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True,
          extra=[a for a in sys.argv[1:] if a.startswith("-") and a != "--force"])
    print(LIB)
