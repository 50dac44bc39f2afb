"""The C-ABI library loads and exports every symbol include/gsx.h declares; struct layouts of
the ctypes mirror match the header.  No compute calls: runs without a GPU."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from intro_to_gaussian_splatting_amd import _ffi

HEADER = os.path.join(ROOT, "include", "gsx.h")


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gsx_[a-z_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    names = _declared_functions()
    for must in ["gsx_version", "gsx_last_error", "gsx_default_params", "gsx_workspace_bytes", "gsx_preprocess",
                 "gsx_render_preprocessed", "gsx_render_forward", "gsx_project_points"]:
        assert must in names


def test_library_exports_every_declared_symbol():
    lib = _ffi.load()
    for name in _declared_functions():
        assert hasattr(lib, name), "libgsx.so does not export %s" % name
        assert name in _ffi.SIGNATURES, "ctypes binding lacks %s" % name
    assert lib.gsx_version() == 100


def test_struct_layouts():
    assert ctypes.sizeof(_ffi.GsxCamera) == 16 * 4 * 2 + 4 * 4 + 2 * 4 + 3 * 4
    assert ctypes.sizeof(_ffi.GsxParams) == 16 * 4 + 8 + 16
    assert ctypes.sizeof(_ffi.GsxFrameStats) == 64
    p = _ffi.default_params()
    assert p.semantics == _ffi.GSX_SEM_REF_CPU and p.layout == _ffi.GSX_LAYOUT_WH3
    assert p.tile_x1 == -1 and p.tile_y1 == -1 and p.out_w == 0


def test_argument_errors_do_not_need_a_gpu():
    lib = _ffi.load()
    assert lib.gsx_workspace_bytes(-1, 64, 64, 16, 10) == 0
    assert lib.gsx_workspace_bytes(10, 0, 64, 16, 10) == 0
    # pure host arithmetic (no library primitive with device queries is left on the path)
    small, big = lib.gsx_workspace_bytes(1000, 256, 256, 16, 8000), lib.gsx_workspace_bytes(1_000_000, 1920, 1080, 16, 5_000_000)
    assert 0 < small < big and big % 256 == 0
    assert big >= 1_000_000 * (16 + 48 + 8 + 8 + 16) + 5_000_000 * 16
    rc = lib.gsx_project_points(None, None, 0, None, None, None)
    assert rc == _ffi.GSX_ERR_INVALID_ARGUMENT
    assert b"camera" in lib.gsx_last_error()
    with pytest.raises(_ffi.GsxError):
        _ffi.check(rc)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_ffi, "_lib", None)
    monkeypatch.setattr(_ffi, "LIB_PATH", str(tmp_path / "libgsx.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _ffi.load()
