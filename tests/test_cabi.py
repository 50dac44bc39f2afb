"""The C-ABI library loads and exports every symbol include/gsx.h declares; struct layouts of
the ctypes mirror match the header.  No compute calls: runs without a GPU."""
import ctypes
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT
from intro_to_gaussian_splatting_amd import _ffi

HEADER = os.path.join(ROOT, "include", "gsx.h")


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gsx_[a-z0-9_]+)\s*\(", text)))


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
    assert lib.gsx_version() == 305


def _exported(path):
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted(line.split()[-1] for line in out.splitlines() if line.strip())


@pytest.mark.skipif(shutil.which("nm") is None, reason="needs binutils nm")
def test_shipping_library_exports_exactly_the_header():
    """libgsx.so is built with -fvisibility=hidden: its dynamic symbol table holds the functions include/gsx.h
    declares and nothing else of ours -- no test hook, no internal C++ symbol (the __hip_cuid_* markers are
    emitted by hipcc for every translation unit).  The hooks live in libgsx_test.so (csrc/gsx_debug.h)."""
    names = [n for n in _exported(_ffi.LIB_PATH) if not n.startswith("__hip_")]
    assert names == _declared_functions(), names
    test_names = [n for n in _exported(_ffi.TEST_LIB_PATH) if not n.startswith("__hip_")]
    assert sorted(set(test_names) - set(names)) == sorted(_ffi.DEBUG_SIGNATURES), test_names


def test_shipping_library_reads_no_environment_variable():
    """The measurement knobs (GSX_DEPTH_SORT, GSX_TILE_SCHEDULE, GSX_BLEND_VARIANT, ...) exist only under
    -DGSX_TEST_HOOKS: the product library does not import getenv at all."""
    if shutil.which("nm") is None:
        pytest.skip("needs binutils nm")
    und = subprocess.run(["nm", "-D", "--undefined-only", _ffi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in und
    und_test = subprocess.run(["nm", "-D", "--undefined-only", _ffi.TEST_LIB_PATH], capture_output=True, text=True,
                              check=True).stdout
    assert "getenv" in und_test


def test_python_constants_equal_the_headers():
    """Every GSX_ERR_* / GSX_SEM_* / GSX_LAYOUT_* / GSX_FLAG_* the ctypes binding spells out is the header's value, and the
    binding knows every flag the header defines."""
    import re
    from intro_to_gaussian_splatting_amd import _ffi

    text = open(HEADER).read()
    defines = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"^#define\s+(GSX_[A-Z0-9_]+)\s+(-?(?:0x)?[0-9a-fA-F]+)\b", text, re.M)}
    enums = {m.group(1): int(m.group(2)) for m in re.finditer(r"\b(GSX_(?:OK|ERR|SEM|LAYOUT)[A-Z0-9_]*)\s*=\s*(-?\d+)", text)}
    known = {**defines, **enums}
    mine = {k: v for k, v in vars(_ffi).items() if re.match(r"GSX_(OK|ERR|SEM|LAYOUT|FLAG)", k) and isinstance(v, int)}
    assert mine, "no constants found in _ffi"
    for k, v in mine.items():
        assert k in known, "%s is not in include/gsx.h" % k
        assert known[k] == v, (k, v, known[k])
    assert {k for k in defines if k.startswith("GSX_FLAG_")} <= set(mine), sorted(set(defines) - set(mine))


def test_struct_layouts():
    assert ctypes.sizeof(_ffi.GsxCamera) == 16 * 4 * 2 + 4 * 4 + 2 * 4 + 3 * 4
    assert ctypes.sizeof(_ffi.GsxParams) == 16 * 4 + 8 + 16 + 8 + 8 + 24 + 8 + 8 + 8 + 8 and _ffi.GsxParams.kept_hint.offset == 88
    assert _ffi.GsxParams.stats_size.offset == 128 and _ffi.GsxParams.original_index.offset == 136 and _ffi.GsxParams.block_bounds.offset == 144 and _ffi.GsxParams.row_of_index.offset == 152   # (128 = the struct of ABI 302 .. 304)
    assert _ffi.GsxParams.hints.offset == 96 and _ffi.GsxParams.n_substrips.offset == 104       # (104 = the ABI-300 struct)
    assert _ffi.GsxParams.substrip_bounds.offset == 112 and _ffi.GsxParams.substrip_events.offset == 120
    assert _ffi.GsxFrameStats.n_kept.offset == 56 and _ffi.GsxFrameStats.stage_ms.offset == 32
    assert ctypes.sizeof(_ffi.GsxFrameStats) == 72 and _ffi.GsxFrameStats.n_redo.offset == 64
    p = _ffi.default_params()
    assert p.semantics == _ffi.GSX_SEM_REF_CPU and p.layout == _ffi.GSX_LAYOUT_WH3
    assert p.tile_x1 == -1 and p.tile_y1 == -1 and p.out_w == 0
    assert p.stats_size == ctypes.sizeof(_ffi.GsxFrameStats) == 72


def test_stats_size_is_validated_and_plain_footprints_needs_n_redo():
    """GsxParams.stats_size (ABI 305): 0 / 64 (the ABI-300 struct) and >= 72 are accepted, anything else is refused before
    anything runs; GSX_FLAG_PLAIN_FOOTPRINTS is refused when n_redo could not reach the caller (a 64-byte struct, a
    GsxParams that ends before the field).  (That the 64-byte struct is not overrun: test_hip_parity, with a canary.)"""
    lib = _ffi.load()
    cam = _ffi.GsxCamera()
    cam.width, cam.height = 64, 64
    out = ctypes.c_void_p(256)          # never dereferenced: the workspace check fails first
    st = _ffi.GsxFrameStats()
    args = lambda par: (ctypes.byref(cam), None, None, None, None, None, 0, 16, out, ctypes.byref(par), ctypes.byref(st),  # noqa: E731
                        None, 0, None)
    for size, ok in ((0, True), (64, True), (72, True), (80, True), (68, False), (56, False), (-8, False), (71, False)):
        p = _ffi.default_params()
        p.stats_size = size
        rc = lib.gsx_render_forward(*args(p))
        assert rc == _ffi.GSX_ERR_INVALID_ARGUMENT
        assert (b"workspace" in lib.gsx_last_error()) == ok and (b"stats_size" in lib.gsx_last_error()) == (not ok), size
    for shrink in ("stats_size", "struct_size"):
        p = _ffi.default_params()
        p.flags |= _ffi.GSX_FLAG_PLAIN_FOOTPRINTS
        assert lib.gsx_render_forward(*args(p)) == _ffi.GSX_ERR_INVALID_ARGUMENT and b"workspace" in lib.gsx_last_error()
        if shrink == "stats_size":
            p.stats_size = 64
        else:
            p.struct_size = 128         # the struct of ABI 302 .. 304: no stats_size
        assert lib.gsx_render_forward(*args(p)) == _ffi.GSX_ERR_INVALID_ARGUMENT and b"PLAIN_FOOTPRINTS" in lib.gsx_last_error()


def test_argument_errors_do_not_need_a_gpu():
    lib = _ffi.load()
    assert lib.gsx_workspace_bytes(-1, 64, 64, 16, 10) == 0
    assert lib.gsx_workspace_bytes(10, 0, 64, 16, 10) == 0
    # pure host arithmetic (no library primitive with device queries is left on the path)
    small, big = lib.gsx_workspace_bytes(1000, 256, 256, 16, 8000), lib.gsx_workspace_bytes(1_000_000, 1920, 1080, 16, 5_000_000)
    assert 0 < small < big and big % 256 == 0
    assert lib.gsx_hints_bytes(0, 64, 16) == 0
    # header + 256 splitters + 2048 samples + list lengths (120 x 68 tiles at 1080p) + per-XCD schedule (tiles + tiles / 32 + 64)
    assert lib.gsx_hints_bytes(1920, 1080, 16) == 256 + 1024 + 8192 + 32768 + 34048
    # (round 3 sized the schedule by the frame's LONGER axis and a frame of more than ~512 tiles along the shorter one
    # overran it; the bound itself is swept in tests/host/plan_sanitize.cpp)
    t = 625 * 625
    assert lib.gsx_hints_bytes(10000, 10000, 16) >= 256 + 1024 + 8192 + 4 * t + 4 * (t + t // 32 + 8)
    assert big >= 1_000_000 * (16 + 48 + 8 + 8 + 16) + 5_000_000 * 16
    rc = lib.gsx_project_points(None, None, 0, None, None, None)
    assert rc == _ffi.GSX_ERR_INVALID_ARGUMENT
    assert b"camera" in lib.gsx_last_error()
    with pytest.raises(_ffi.GsxError):
        _ffi.check(rc)


def test_params_struct_size_is_checked_before_anything_runs():
    """GsxParams.struct_size (include/gsx.h): gsx_default_params states the size this build knows; a size the library
    does not know is refused; a struct that ends before `hints` (a client built against an older header) has that
    field ignored -- here: a misaligned pointer in it is not even looked at."""
    lib = _ffi.load()
    p = _ffi.default_params()
    assert p.struct_size == ctypes.sizeof(_ffi.GsxParams)
    cam = _ffi.GsxCamera()
    cam.width, cam.height = 64, 64
    out = ctypes.c_void_p(256)          # never dereferenced: the workspace check fails first
    args = lambda par: (ctypes.byref(cam), None, None, None, None, None, 0, 16, out, ctypes.byref(par), None, None, 0, None)  # noqa: E731
    p.struct_size = 12
    assert lib.gsx_render_forward(*args(p)) == _ffi.GSX_ERR_INVALID_ARGUMENT and b"struct_size" in lib.gsx_last_error()
    p = _ffi.default_params()
    p.hints = 257                       # misaligned: refused when the field counts ...
    assert lib.gsx_render_forward(*args(p)) == _ffi.GSX_ERR_INVALID_ARGUMENT and b"hints" in lib.gsx_last_error()
    p.struct_size = _ffi.GsxParams.hints.offset     # ... and not read when the caller's struct ends before it
    assert lib.gsx_render_forward(*args(p)) == _ffi.GSX_ERR_INVALID_ARGUMENT and b"workspace" in lib.gsx_last_error()


def test_default_params_never_writes_behind_the_callers_struct():
    """A client compiled against the 104-byte ABI-300 struct: gsx_default_params_sized(p, 104) -- and the exported
    legacy function such a binary calls -- fill 104 bytes, state struct_size = 104 and leave the canary behind the
    struct alone; a later call then reads nothing behind it either (n_substrips there would be refused)."""
    lib = _ffi.load()
    for call in (lambda b: lib.gsx_default_params_sized(ctypes.cast(b, ctypes.POINTER(_ffi.GsxParams)), 104),
                 lambda b: lib.gsx_default_params(ctypes.cast(b, ctypes.POINTER(_ffi.GsxParams)))):
        buf = (ctypes.c_ubyte * 160)(*([0xAB] * 160))
        call(buf)
        assert bytes(buf[104:]) == b"\xab" * 56, "wrote behind a 104-byte struct"
        p = ctypes.cast(buf, ctypes.POINTER(_ffi.GsxParams)).contents
        assert p.struct_size == 104 and p.tile_x1 == -1 and p.semantics == _ffi.GSX_SEM_REF_CPU and p.hints is None
        # 0xABABABAB in n_substrips (offset 104) would be refused if it were read: the call gets as far as the workspace check
        cam = _ffi.GsxCamera()
        cam.width, cam.height = 64, 64
        rc = lib.gsx_render_forward(ctypes.byref(cam), None, None, None, None, None, 0, 16, ctypes.c_void_p(256), p, None,
                                    None, 0, None)
        assert rc == _ffi.GSX_ERR_INVALID_ARGUMENT and b"workspace" in lib.gsx_last_error()
    # sizes this library does not know are clamped to what it has / to the smallest struct there ever was
    buf = (ctypes.c_ubyte * 256)(*([0xAB] * 256))
    lib.gsx_default_params_sized(ctypes.cast(buf, ctypes.POINTER(_ffi.GsxParams)), 200)
    assert ctypes.cast(buf, ctypes.POINTER(_ffi.GsxParams)).contents.struct_size == ctypes.sizeof(_ffi.GsxParams)
    assert bytes(buf[ctypes.sizeof(_ffi.GsxParams):]) == b"\xab" * (256 - ctypes.sizeof(_ffi.GsxParams))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_ffi, "_lib", None)
    monkeypatch.setattr(_ffi, "LIB_PATH", str(tmp_path / "libgsx.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _ffi.load()


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_a_c99_client_links_and_runs(tmp_path):
    """include/gsx.h is plain C: a C99 client (tests/host/cabi_smoke.c, -Wall -Wextra -pedantic) compiles against it, links
    libgsx.so and exercises what needs no GPU -- the version, the defaults at the sizes IT compiled (canary behind the
    struct), the size arithmetic and two argument errors -- the way a maintainer's C / C++ viewer would."""
    exe = str(tmp_path / "cabi_smoke")
    lib_dir = os.path.dirname(_ffi.LIB_PATH)
    build = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                            os.path.join(ROOT, "tests", "host", "cabi_smoke.c"), "-o", exe, _ffi.LIB_PATH,
                            "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "GsxParams %d bytes, GsxFrameStats %d bytes" % (ctypes.sizeof(_ffi.GsxParams), ctypes.sizeof(_ffi.GsxFrameStats)) in run.stdout


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_the_c_snippets_of_the_integration_document_compile(tmp_path):
    """The two C loops of INTEGRATION.md (a viewer's frame loop with the hand-over between frames and the n_redo check; a
    multi-GPU rank compositing its strip in parts) are type-checked against include/gsx.h: every field, flag and call they
    use exists with that type.  (HIP's own names are declared by a five-line prelude: the snippets are about gsx.)"""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```c\n(.*?)```", text, flags=re.S)
    assert len(blocks) == 2
    prelude = """
#include <stdlib.h>
#include "gsx.h"
typedef void *hipEvent_t; typedef void *hipStream_t;
enum { hipEventDisableTiming = 2 };
int hipMalloc(void **p, size_t n); int hipMemsetAsync(void *p, int v, size_t n, hipStream_t s);
int hipEventCreateWithFlags(hipEvent_t *e, unsigned flags); int hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
extern GsxCamera cam; extern const float *means, *scales, *quats, *opacity, *colors; extern int64_t n; extern float *image, *strip;
extern void *ws; extern size_t ws_bytes; extern hipStream_t stream, comm_stream; extern int32_t col0, col1;
"""
    src = prelude + "void viewer(void) {\n" + blocks[0] + "\n}\nvoid rank(void) {\nGsxParams prm; GsxFrameStats st; gsx_default_params(&prm);\n" + blocks[1] + "\n(void)st;\n}\n"
    path = tmp_path / "snippets.c"
    path.write_text(src)
    out = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wno-unused-variable", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(path)],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
