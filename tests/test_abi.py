"""The C-ABI library loads (no GPU needed) and exports every symbol include/tpose_hip.h declares."""
import os
import re

from tpose_amd import build as tb
from tpose_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if fn.endswith(".h"):
            text = open(os.path.join(ROOT, "include", fn)).read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            names += re.findall(r"\b(tp_[a-z_0-9]+)\s*\(", text)
    return sorted(set(names))


def test_library_builds_and_exports_every_declared_symbol():
    tb.build()
    lib = capi.load()
    decl = declared_symbols()
    assert len(decl) >= 20
    for name in decl:
        assert hasattr(lib, name), "libtpose_hip.so does not export %s" % name
    assert sorted(capi.SYMBOLS) == decl
    assert lib.tp_abi_version() == 5


def test_no_device_fails_loudly():
    """Without a GPU the product path must refuse to run (no CPU fallback)."""
    if capi.device_count() > 0:
        return
    try:
        capi.Context(0, 64, 64)
    except capi.TposeError as e:
        assert e.code == capi.TP_ERR_NO_DEVICE
    else:
        raise AssertionError("tp_create succeeded without a device")


def test_product_never_touches_the_oracle():
    """Nothing under tpose_amd/ or include/ may reference oracle/."""
    for base in ("tpose_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hpp", ".hip", ".cpp", ".c")) or f == "Makefile":
                    text = open(os.path.join(dp, f), errors="ignore").read()
                    hit = re.search(r"(from|import)\s+oracle|oracle[/.]|tp_oracle|tpo_[a-z]", text)
                    assert hit is None, "%s references the oracle (%s)" % (os.path.join(dp, f), hit.group(0))
