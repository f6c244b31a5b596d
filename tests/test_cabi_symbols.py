"""CPU: the C-ABI shared library loads (no GPU needed) and exports every symbol declared in
include/dcs_hip.h with the ctypes signature table of dcs_amd/lib.py; no compute calls."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "dcs_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|const char\*)\s+(dcs_\w+)\s*\(", txt)))


def test_header_and_binding_table_agree():
    from dcs_amd import lib
    names = declared_symbols()
    assert len(names) >= 30
    assert sorted(list(lib.SIGNATURES) + ["dcs_version"]) == names


def test_library_loads_and_exports_everything():
    from dcs_amd import lib
    if not os.path.exists(lib.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    l = lib.load()
    for name in declared_symbols():
        assert hasattr(l, name), name
    assert b"gfx950" in l.dcs_version()


def test_argument_errors_are_reported_not_crashed():
    """Every launcher validates its arguments before touching the GPU: null pointers -> DCS_E_ARG."""
    from dcs_amd import lib
    l = lib.load()
    assert l.dcs_reduce_slab(None, None, 0, 0, 0, 0, 0, None) == -1
    assert l.dcs_bn_act(None, None, None, None, None, 0, 0, 0, None) == -1
    g = lib.DcsConvGeom()
    assert l.dcs_conv_gather(None, None, None, None, g, 0, None, None) == -1
    with pytest.raises(RuntimeError, match="DCS_E_ARG"):
        lib.check(-1, "x")


def test_missing_library_fails_loudly(monkeypatch):
    from dcs_amd import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/libdcs_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lib.load()
