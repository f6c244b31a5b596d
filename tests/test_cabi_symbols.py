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
    assert l.dcs_bn_act(None, None, None, None, None, 0, 0, 0, None, None) == -1
    g = lib.DcsConvGeom()
    assert l.dcs_conv_gather(None, None, None, None, g, 0, None, None) == -1
    with pytest.raises(RuntimeError, match="DCS_E_ARG"):
        lib.check(-1, "x")


def test_library_switches_are_set_through_the_abi_not_the_environment(monkeypatch):
    """The launchers never read the environment: the DCS_* variables are read once at load time and tests / tools flip a
    switch through dcs_set_option.  Unknown names are argument errors."""
    from dcs_amd import lib
    l = lib.load()
    assert lib.get_option("bn_nt") in (0, 1) and lib.get_option("nt_min_mb") > 0
    old = lib.set_option("x3_halo", 2)
    try:
        monkeypatch.setenv("DCS_X3_HALO", "0")                 # changing the environment after load has no effect
        assert lib.get_option("x3_halo") == 2
    finally:
        lib.set_option("x3_halo", old)
    assert lib.get_option("x3_halo") == old
    assert l.dcs_set_option(b"no_such_switch", 1) == -1 and l.dcs_set_option(None, 1) == -1
    src = os.path.join(ROOT, "doubly-contrastive-semseg_amd", "dcs_amd", "csrc")
    for f in os.listdir(src):
        if f.endswith((".hip", ".h")):
            assert "getenv" not in open(os.path.join(src, f)).read(), f


def test_missing_library_fails_loudly(monkeypatch):
    from dcs_amd import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/libdcs_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lib.load()


def test_conv_gather_refuses_windows_beyond_32bit_addressing():
    """conv_gather_kernel addresses its source relative to the first image of a block and the weights from their base
    with 32-bit byte offsets (hardware range-checked buffer loads).  A geometry whose two-image window or weight tensor
    exceeds 2 GiB - 1 must come back as DCS_E_UNSUPPORTED from the launcher -- the check precedes the launch, so no GPU
    (and no allocation: the pointers are never dereferenced) is needed.  Mirrors the check in dcs_conv_wgrad."""
    import ctypes as C
    from dcs_amd import lib, ops
    l = lib.load()
    fake = C.c_void_p(0x100000)                         # 16-byte aligned, never dereferenced
    # 1x1 convolution on 2 images of 2048 channels x 600 x 600: one image = 2.95 GB
    big = ops.geom_fwd(2, 600, 600, 2048, 64, 1, 1, 1, 0)
    assert l.dcs_conv_gather(fake, fake, None, fake, C.byref(big), 0, None, None) == -3
    assert l.dcs_conv_gather_split(fake, fake, fake, C.byref(big), 2, 2 * 600 * 600 * 64, None) == -3
    # one image fits (1.47 GB) but a block's 128 pixels straddle two images only if N > 1
    one = ops.geom_fwd(1, 424, 424, 2048, 64, 1, 1, 1, 0)
    two = ops.geom_fwd(2, 424, 424, 2048, 64, 1, 1, 1, 0)
    assert l.dcs_conv_gather(fake, fake, None, fake, C.byref(two), 0, None, None) == -3
    assert l.dcs_conv_gather(fake, fake, None, fake, C.byref(one), 0, None, None) != -3   # passes the check (-2 w/o GPU)
    # weights beyond 2 GiB: 16384 -> 16384 channels 3x3
    wide = ops.geom_fwd(1, 8, 8, 16384, 16384, 3, 3, 1, 1)
    assert l.dcs_conv_gather(fake, fake, None, fake, C.byref(wide), 0, None, None) == -3
    with pytest.raises(RuntimeError, match="DCS_E_UNSUPPORTED"):
        lib.check(-3, "dcs_conv_gather")
