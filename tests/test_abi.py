"""The C-ABI library loads without a GPU and exports every symbol include/hirl4ucav.h (product) and include/hirl4ucav_debug.h
(measurement helpers) declare — and nothing else named hx_* (no compute)."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(headers=("hirl4ucav.h", "hirl4ucav_debug.h")):
    out = set()
    for h in headers:
        text = open(os.path.join(REPO, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        out |= set(re.findall(r"\b(hx_[a-z0-9_]+)\s*\(", text))
    return sorted(out)


def test_header_declares_the_hot_path():
    syms = declared_symbols()
    for must in ("hx_env_step", "hx_env_reset", "hx_label_transitions", "hx_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    so = os.path.join(REPO, "hirl4ucav_amd", "libhx_mi355.so")
    if not os.path.exists(so):
        import __graft_entry__ as g

        g.build()
    L = ctypes.CDLL(so)
    missing = [s for s in declared_symbols() if not hasattr(L, s)]
    assert not missing, missing
    L.hx_version.restype = ctypes.c_int
    assert L.hx_version() >= 100
    # ... and exports nothing the headers do not declare; the debug helpers live in their own header
    import subprocess

    nm = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True).stdout
    exported = sorted({ln.split()[-1] for ln in nm.splitlines() if ln.split() and ln.split()[-1].startswith("hx_") and " T " in ln})
    assert exported == declared_symbols(), sorted(set(exported) ^ set(declared_symbols()))
    assert not [s for s in declared_symbols(("hirl4ucav.h",)) if s.startswith("hx_debug")]


def test_argument_errors_are_reported_not_swallowed():
    from hirl4ucav_amd import _lib

    with pytest.raises(_lib.HxError, match="hx_env_step"):
        _lib.call("hx_env_step", None, 0, 0, None, None, None, None, None, None, None)


def test_product_does_not_import_the_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    bad = []
    for root, _, files in os.walk(os.path.join(REPO, "hirl4ucav_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                if "oracle" in open(os.path.join(root, f), errors="ignore").read().replace("the scalar oracle", ""):
                    bad.append(os.path.join(root, f))
    assert not bad, bad


def test_abi_version_and_struct_sizes_match_the_binding():
    """HX_ABI_VERSION of the header == hx_version() of the library == the binding's, and every struct that crosses the boundary has the size
    the library was compiled with (hx_abi_sizes) — the check _lib.load() / the engines run at construction (a binding that had
    drifted from the header handed the kernels a 9-word statistics buffer under a library that adds into 512 words)."""
    import re

    from hirl4ucav_amd import _lib
    from hirl4ucav_amd.agents import engine as E
    from hirl4ucav_amd.agents import sac_engine as SE

    hdr = open(os.path.join(REPO, "include", "hirl4ucav.h")).read()
    declared = int(re.search(r"#define HX_ABI_VERSION (\d+)", hdr).group(1))
    L = _lib.load()
    assert declared == _lib.ABI_VERSION == L.hx_version()
    sizes = (ctypes.c_int32 * 8)()
    L.hx_abi_sizes.argtypes = [ctypes.POINTER(ctypes.c_int32)]
    assert L.hx_abi_sizes(sizes) == 0
    for i, cls in enumerate((_lib.HxStepOpts, E.HxNets, E.HxHyper, E.HxBatch, E.HxSample, SE.HxSacNets, SE.HxSacBatch)):
        assert sizes[i] == ctypes.sizeof(cls), (cls.__name__, sizes[i], ctypes.sizeof(cls))
        _lib.check_struct(i, cls)
    assert sizes[7] == _lib.STAT_WAYS * _lib.STAT_PITCH

    class Short(ctypes.Structure):
        _fields_ = [("a", ctypes.c_void_p)]

    with pytest.raises(_lib.HxError, match="ABI mismatch"):
        _lib.check_struct(1, Short)
