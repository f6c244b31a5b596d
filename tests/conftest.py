import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "doubly-contrastive-semseg_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture()
def libopt(monkeypatch):
    """Set a switch of libdcs_hip.so for one test: the library reads its DCS_* environment variables once at load time, so
    a test changes them through dcs_set_option (and mirrors the value into the environment for the Python-side readers of
    the same name); restored afterwards."""
    from dcs_amd import lib
    saved = {}

    def set_(name, value):
        old = lib.set_option(name, int(value))
        saved.setdefault(name, old)
        monkeypatch.setenv("DCS_" + name.upper(), str(value))
    yield set_
    for name, old in saved.items():
        lib.set_option(name, old)
