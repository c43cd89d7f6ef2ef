"""CPU: the C-ABI library builds, loads and exports every symbol that
include/spherehand_hip.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "spherehand_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(shr_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_hot_path():
    syms = declared_symbols()
    for s in ("shr_sphere_raster_fwd", "shr_sphere_raster_bwd", "shr_abi_version", "shr_error_string"):
        assert s in syms


def test_library_builds_and_exports_every_declared_symbol():
    from spherehand_amd import build
    so = build.build()
    h = ctypes.CDLL(so)
    for s in declared_symbols():
        assert hasattr(h, s), "libspherehand_hip.so does not export %s" % s


def test_loader_signatures_cover_the_header():
    from spherehand_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    h = _lib.lib()
    assert h.shr_abi_version() == _lib.ABI_VERSION
    assert h.shr_error_string(0) == b"ok"
    assert b"invalid" in h.shr_error_string(-1)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from spherehand_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "SO_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.SphereHandLibraryError):
        _lib.lib()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "spherehand_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("# oracle-free", ""), \
                    "%s mentions the oracle" % os.path.join(dirpath, f)
    txt = open(os.path.join(ROOT, "depth_rasterization.py")).read() if os.path.exists(
        os.path.join(ROOT, "depth_rasterization.py")) else ""
    assert "oracle" not in txt
