"""CPU tests of the C-ABI boundary: the library loads without a GPU and exports exactly the
symbols include/selavi_hip.h declares (no compute calls here)."""
import ctypes
import os

import pytest

from selavi_amd import _lib


@pytest.fixture(scope="module")
def lib():
    from selavi_amd import build
    build.build(verbose=False)
    return ctypes.CDLL(_lib.LIBPATH)


def test_header_parses():
    d = _lib.parse_header()
    assert "slv_sk_pass" in d and "slv_version" in d
    ret, args = d["slv_sk_pass"]
    assert ret == "int" and [a[1] for a in args][:3] == ["P", "N_local", "N_global"]


def test_library_exports_every_declared_symbol(lib):
    for name in _lib.declared_symbols():
        assert hasattr(lib, name), f"{name} declared in include/selavi_hip.h but not exported"


def test_version_and_error_string(lib):
    L = _lib.load()
    assert L.slv_version() >= 1
    assert isinstance(L.slv_last_error(), (bytes, type(None)))
    assert L.slv_sk_workspace_bytes(309, 512) > 512 * 320 * 8


def test_no_cpu_fallback_message(monkeypatch):
    monkeypatch.setattr(_lib, "LIBPATH", "/nonexistent/libselavi_hip.so")
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(_lib.SelaviHipError):
        _lib.load()
