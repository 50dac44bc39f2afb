"""ctypes front end of oracle/liboracle.so (the C restatement; test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

from .cpu_ref import Camera, Preprocessed

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


class _OrcCamera(ctypes.Structure):
    _fields_ = [("V", ctypes.c_float * 16), ("F", ctypes.c_float * 16),
                ("tan_fovx", ctypes.c_float), ("tan_fovy", ctypes.c_float),
                ("fx", ctypes.c_float), ("fy", ctypes.c_float),
                ("width", ctypes.c_int32), ("height", ctypes.c_int32)]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "raster_cpu.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B", "liboracle.so"], check=True, capture_output=True)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_preprocess.restype = ctypes.c_int
        _lib.orc_render.restype = ctypes.c_int
    return _lib


def _fp(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _cam(cam: Camera) -> _OrcCamera:
    c = _OrcCamera()
    c.V[:] = np.asarray(cam.world2view, np.float32).reshape(-1).tolist()
    c.F[:] = np.asarray(cam.full_proj, np.float32).reshape(-1).tolist()
    c.tan_fovx, c.tan_fovy = float(cam.tan_fovx), float(cam.tan_fovy)
    c.fx, c.fy = float(cam.fx), float(cam.fy)
    c.width, c.height = int(cam.width), int(cam.height)
    return c


def preprocess(points, colors, scales, quats, opacity_logit, cam: Camera) -> Preprocessed:
    f = lambda a, w: np.ascontiguousarray(np.asarray(a, np.float32).reshape(-1, w))  # noqa: E731
    points, colors, scales, quats = f(points, 3), f(colors, 3), f(scales, 3), f(quats, 4)
    op = f(opacity_logit, 1)
    n = points.shape[0]
    z = lambda *s: np.zeros(s, np.float32)  # noqa: E731
    xy, col, c2, dep, inv, rad = z(n, 2), z(n, 3), z(n, 2, 2), z(n), z(n, 2, 2), z(n)
    mnx, mxx, mny, mxy, sop = z(n), z(n), z(n), z(n), z(n, 1)
    order = np.zeros(n, np.int64)
    nvis = ctypes.c_int64(0)
    c = _cam(cam)
    rc = lib().orc_preprocess(ctypes.byref(c), _fp(points), _fp(colors), _fp(scales), _fp(quats), _fp(op),
                              ctypes.c_int64(n), _fp(xy), _fp(col), _fp(c2), _fp(dep), _fp(inv), _fp(rad),
                              _fp(mnx), _fp(mxx), _fp(mny), _fp(mxy), _fp(sop),
                              order.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), ctypes.byref(nvis))
    assert rc == 0
    m = nvis.value
    return Preprocessed(points=xy[:m], colors=col[:m], covariance_2d=c2[:m], depths=dep[:m],
                        inverse_covariance_2d=inv[:m], radius=rad[:m], points_xy=xy[:m],
                        min_x=mnx[:m], min_y=mny[:m], max_x=mxx[:m], max_y=mxy[:m],
                        sigmoid_opacity=sop[:m], order=order[:m])


def render(pre, width: int, height: int, tile: int = 16, nthreads: Optional[int] = None,
           window: Optional[Tuple[int, int, int, int]] = None, exact: bool = False):
    """Returns (image (W,H,3) indexed [x,y], pairs, instances).  ``exact``: the same rules evaluated in
    float64 from the same float32 stage-1 arrays (the exact-arithmetic limit of the reference's formulas, not
    its float32 behaviour): tells which side is off when kernel and restatement disagree."""
    f = lambda a: np.ascontiguousarray(np.asarray(a, np.float32))  # noqa: E731
    means, colors, inv = f(pre.points), f(pre.colors), f(pre.inverse_covariance_2d)
    mnx, mxx, mny, mxy, sop = f(pre.min_x), f(pre.max_x), f(pre.min_y), f(pre.max_y), f(pre.sigmoid_opacity)
    n = means.shape[0]
    image = np.zeros((width, height, 3), np.float32)
    pairs, inst = ctypes.c_int64(0), ctypes.c_int64(0)
    win = None
    if window is not None:
        win = (ctypes.c_int32 * 4)(*[int(v) for v in window])
    if nthreads is None:
        nthreads = os.cpu_count() or 1
    lib().orc_set_exact(1 if exact else 0)
    try:
        rc = lib().orc_render(int(height), int(width), int(tile), _fp(means), _fp(colors), _fp(inv),
                              _fp(mnx), _fp(mxx), _fp(mny), _fp(mxy), _fp(sop), ctypes.c_int64(n), _fp(image),
                              int(nthreads), win, ctypes.byref(pairs), ctypes.byref(inst))
    finally:
        lib().orc_set_exact(0)
    assert rc == 0
    return image, pairs.value, inst.value


def render_cuda_semantics(pre, width: int, height: int, nthreads: Optional[int] = None) -> np.ndarray:
    """The reference's CUDA-kernel semantics (splat/c/render.cu) on the CPU; returns (H,W,3)."""
    f = lambda a: np.ascontiguousarray(np.asarray(a, np.float32))  # noqa: E731
    means, colors, inv = f(pre.points), f(pre.colors), f(pre.inverse_covariance_2d)
    mnx, mxx, mny, mxy, sop = f(pre.min_x), f(pre.max_x), f(pre.min_y), f(pre.max_y), f(pre.sigmoid_opacity)
    image = np.zeros((height, width, 3), np.float32)
    if nthreads is None:
        nthreads = os.cpu_count() or 1
    fn = lib().orc_render_cuda_semantics
    fn.restype = ctypes.c_int
    rc = fn(int(height), int(width), _fp(means), _fp(colors), _fp(inv), _fp(mnx), _fp(mxx), _fp(mny), _fp(mxy),
            _fp(sop), ctypes.c_int64(means.shape[0]), _fp(image), int(nthreads))
    assert rc == 0
    return image


def render_std3dgs(points, colors, scales, quats, opacity_logit, cam: Camera, tile: int = 16,
                   background=(0.0, 0.0, 0.0), nthreads: Optional[int] = None,
                   window: Optional[Tuple[int, int, int, int]] = None):
    """GSX_SEM_STD_3DGS on the CPU (orc_render_std3dgs; parity unpinned, see raster_cpu.c).
    Returns (image (H,W,3) indexed [y,x], n_visible, instances, stage1 (n,8))."""
    f = lambda a, w: np.ascontiguousarray(np.asarray(a, np.float32).reshape(-1, w))  # noqa: E731
    points, colors, scales, quats = f(points, 3), f(colors, 3), f(scales, 3), f(quats, 4)
    op = f(opacity_logit, 1)
    n = points.shape[0]
    image = np.zeros((int(cam.height), int(cam.width), 3), np.float32)
    stage1 = np.zeros((max(n, 1), 8), np.float32)
    bg = (ctypes.c_float * 3)(*[float(v) for v in background])
    win = None
    if window is not None:
        win = (ctypes.c_int32 * 4)(*[int(v) for v in window])
    if nthreads is None:
        nthreads = os.cpu_count() or 1
    nvis, inst = ctypes.c_int64(0), ctypes.c_int64(0)
    c = _cam(cam)
    fn = lib().orc_render_std3dgs
    fn.restype = ctypes.c_int
    rc = fn(ctypes.byref(c), _fp(points), _fp(colors), _fp(scales), _fp(quats), _fp(op), ctypes.c_int64(n),
            int(tile), bg, _fp(image), int(nthreads), win, ctypes.byref(nvis), ctypes.byref(inst), _fp(stage1))
    assert rc == 0
    return image, nvis.value, inst.value, stage1[:n]
