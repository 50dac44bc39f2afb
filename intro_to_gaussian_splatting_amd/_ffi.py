"""ctypes binding of libgsx.so (the C ABI declared in include/gsx.h).

The library is built ahead of time by ``__graft_entry__.build()`` /
``make -C intro_to_gaussian_splatting_amd/csrc`` and lives next to this file.  There is no CPU
fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_float, c_int32, c_int64, c_size_t, c_uint8, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgsx.so")
# The same sources built with -DGSX_TEST_HOOKS plus csrc/gsx_debug.hip (csrc/gsx_debug.h): measurement knobs and the
# sorts on caller-provided data.  tests/ and tools/ only -- load_test_hooks(); the product never loads it.
TEST_LIB_PATH = os.path.join(_HERE, "libgsx_test.so")

GSX_OK = 0
GSX_ERR_INVALID_ARGUMENT = -1
GSX_ERR_WORKSPACE_TOO_SMALL = -2
GSX_ERR_HIP = -3
GSX_ERR_UNSUPPORTED = -4

GSX_SEM_REF_CPU = 0
GSX_SEM_REF_CUDA = 1
GSX_SEM_STD_3DGS = 2
GSX_LAYOUT_WH3 = 0
GSX_LAYOUT_HW3 = 1
GSX_FLAG_TIMING = 1
GSX_FLAG_NO_SYNC = 2
GSX_FLAG_GENERIC_KERNELS = 4
GSX_FLAG_PUBLISHED_RECTS = 8
GSX_FLAG_NO_LONG_TILE_SPLIT = 16
GSX_FLAG_TILE_SCHEDULE = 32
GSX_FLAG_NO_TILE_SCHEDULE = 64
GSX_FLAG_HINTS_VALID = 128
GSX_FLAG_SMALL_BATCH = 256
GSX_FLAG_ONE_VISIBLE = 512
GSX_FLAG_PLAIN_FOOTPRINTS = 1024
GSX_BOUNDS_ROWS = 256


def visible_rows_flag(n: int, n_visible: int, flags: int) -> int:
    """include/gsx.h, GSX_FLAG_SMALL_BATCH / GSX_FLAG_ONE_VISIBLE: the flag a call over ``n`` Gaussians has to carry when
    ``n_visible`` of them turned out visible and it was issued with ``flags`` -- 0 when it already assumed the right
    number of rows (from four rows up, or n itself at most three and all of them visible); -1 when it carried one of the
    two flags and four or more Gaussians are visible: issue it again with NEITHER."""
    cls = lambda k: GSX_FLAG_ONE_VISIBLE if k == 1 else (GSX_FLAG_SMALL_BATCH if k <= 3 else 0)  # noqa: E731
    assumed = (flags & (GSX_FLAG_ONE_VISIBLE | GSX_FLAG_SMALL_BATCH)) or cls(n)
    true = cls(n_visible) if n_visible > 0 else assumed
    if true == assumed:
        return 0
    return true if true else -1


def with_rows_flag(flags: int, again: int) -> int:
    """``flags`` with the row class ``visible_rows_flag`` asked for (-1: neither flag)."""
    return (flags & ~(GSX_FLAG_SMALL_BATCH | GSX_FLAG_ONE_VISIBLE)) | max(int(again), 0)
STAGE_NAMES = ("project", "depth_sort", "scan", "bin", "blend", "total")


class GsxCamera(ctypes.Structure):
    _fields_ = [("world2view", c_float * 16), ("full_proj", c_float * 16),
                ("tan_fovx", c_float), ("tan_fovy", c_float), ("fx", c_float), ("fy", c_float),
                ("width", c_int32), ("height", c_int32), ("camera_center", c_float * 3)]


class GsxParams(ctypes.Structure):
    _fields_ = [("semantics", c_int32), ("layout", c_int32),
                ("tile_x0", c_int32), ("tile_x1", c_int32), ("tile_y0", c_int32), ("tile_y1", c_int32),
                ("out_x0", c_int32), ("out_y0", c_int32), ("out_w", c_int32), ("out_h", c_int32),
                ("flags", c_int32), ("background", c_float * 3), ("camera_device", c_void_p),
                ("tile_counts", c_void_p), ("sh", c_void_p), ("sh_degree", c_int32), ("struct_size", c_int32),
                ("kept_hint", c_int64), ("hints", c_void_p),
                ("n_substrips", c_int32), ("substrip_axis", c_int32), ("substrip_bounds", POINTER(c_int32)),
                ("substrip_events", POINTER(c_void_p)), ("stats_size", c_int32), ("reserved1", c_int32),
                ("original_index", c_void_p), ("block_bounds", c_void_p), ("row_of_index", c_void_p)]


class GsxFrameStats(ctypes.Structure):
    _fields_ = [("n_visible", c_int64), ("n_instances", c_int64), ("n_tiles", c_int64), ("reserved", c_int64),
                ("stage_ms", c_float * 6), ("n_kept", c_int64), ("n_redo", c_int64)]


# name -> (restype, argtypes); every symbol include/gsx.h declares.
_FP = c_void_p  # device float*
SIGNATURES = {
    "gsx_version": (ctypes.c_int, []),
    "gsx_last_error": (ctypes.c_char_p, []),
    "gsx_default_params": (None, [POINTER(GsxParams)]),      # (the symbol of ABI 300 / 301 binaries: 104 bytes)
    "gsx_default_params_sized": (None, [POINTER(GsxParams), c_size_t]),
    "gsx_workspace_bytes": (c_size_t, [c_int64, c_int32, c_int32, c_int32, c_int64]),
    "gsx_hints_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "gsx_preprocess": (ctypes.c_int, [POINTER(GsxCamera)] + [_FP] * 5 + [c_int64] + [_FP] * 11 +
                       [c_void_p, POINTER(c_int64), POINTER(GsxParams), c_void_p, c_size_t, c_void_p]),
    "gsx_render_preprocessed": (ctypes.c_int, [c_int32, c_int32, c_int32] + [_FP] * 8 + [c_int64, _FP,
                                POINTER(GsxParams), POINTER(GsxFrameStats), c_void_p, c_size_t, c_void_p]),
    "gsx_render_forward": (ctypes.c_int, [POINTER(GsxCamera)] + [_FP] * 5 + [c_int64, c_int32, _FP,
                           POINTER(GsxParams), POINTER(GsxFrameStats), c_void_p, c_size_t, c_void_p]),
    "gsx_covariance_3d": (ctypes.c_int, [_FP, _FP, c_int64, _FP, c_void_p]),
    "gsx_covariance_2d": (ctypes.c_int, [POINTER(GsxCamera), _FP, _FP, c_int64, _FP, c_void_p]),
    "gsx_project_points": (ctypes.c_int, [POINTER(GsxCamera), _FP, c_int64, _FP, c_void_p, c_void_p]),
    "gsx_sh_to_rgb": (ctypes.c_int, [_FP, _FP, c_int32, c_int64, POINTER(c_float), _FP, c_void_p]),
}

# csrc/gsx_debug.h
DEBUG_SIGNATURES = {
    "gsx_debug_sort_pairs": (ctypes.c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_size_t,
                                            c_void_p]),
    "gsx_debug_set_blend_probe": (ctypes.c_int, [c_void_p]),
    "gsx_debug_depth_sort": (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int32, ctypes.c_uint32,
                                            c_int64, POINTER(c_int64), c_void_p, c_size_t, c_void_p]),
}

_lib = None
_test_lib = None


def _bind(lib, signatures):
    for name, (res, args) in signatures.items():
        fn = getattr(lib, name, None)
        if fn is None and "GSX_TEST_LIB_PATH" in os.environ:     # an older experiment build of the test library
            continue
        if fn is None:
            raise AttributeError("%s is missing from %s" % (name, lib._name))
        fn.restype = res
        fn.argtypes = args
    return lib


def load_test_hooks():
    """libgsx_test.so (tests/ and tools/ only): every entry point of the product library plus the hooks of
    csrc/gsx_debug.h; its kernels additionally honour the GSX_* measurement knobs of the environment."""
    global _test_lib
    if _test_lib is None:
        if not os.path.exists(TEST_LIB_PATH):
            raise RuntimeError("libgsx_test.so not found at %s: build it with `make -C intro_to_gaussian_splatting_amd/csrc`"
                               % TEST_LIB_PATH)
        # GSX_TEST_LIB_PATH: an experiment build of the test library (tools/: A/B of kernel variants on one GPU box)
        _test_lib = _bind(_bind(ctypes.CDLL(os.environ.get("GSX_TEST_LIB_PATH", TEST_LIB_PATH)), SIGNATURES), DEBUG_SIGNATURES)
    return _test_lib


def use_test_library() -> None:
    """Development aid (tools/, bench.py --test-lib): every later load() returns libgsx_test.so, so that a whole
    frame can be rendered under the GSX_* measurement knobs.  Never called by the package itself."""
    global _lib
    _lib = load_test_hooks()


def load():
    """Loads libgsx.so once; raises RuntimeError (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libgsx.so not found at %s: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C intro_to_gaussian_splatting_amd/csrc` (needs hipcc, targets gfx950). "
                "There is no CPU fallback." % LIB_PATH)
        _lib = _bind(ctypes.CDLL(LIB_PATH), SIGNATURES)
    return _lib


class GsxError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__("libgsx error %d: %s" % (code, message))
        self.code = code


def check(code: int) -> None:
    if code != GSX_OK:
        raise GsxError(code, load().gsx_last_error().decode("utf-8", "replace"))


def default_params() -> GsxParams:
    p = GsxParams()
    load().gsx_default_params_sized(ctypes.byref(p), ctypes.sizeof(GsxParams))
    return p
