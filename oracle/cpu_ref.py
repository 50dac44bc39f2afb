"""CPU restatement (numpy float32) of the reference's pure-Python forward rasteriser.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker.  The product path (``intro_to_gaussian_splatting_amd``) never
imports this module and fails loudly when its HIP library is missing.

Parity status: PINNED.  ``oracle/capture_golden.py`` imports the reference itself
(``/root/reference/splat``) in the build container and stores its inputs, every
``PreprocessedScene`` field, the depth permutation and the rendered image under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks this restatement against them.

Every function cites the reference lines it restates (paths relative to /root/reference).
All arithmetic is float32 in the operation order torch EXECUTES for the reference's
expressions (``oracle/probe_torch_order.py``): the products torch folds into one sgemm --
``[p,1] @ M``, ``(N,3,3) @ (3,3)``, ``(1,2) @ (2,2)`` -- are sequential fused multiply-add chains
(``fma`` below, exact), batched 3x3 products and the final dot of the weight round every product
and sum.  The C restatement (``oracle/raster_cpu.c``) and the HIP projection kernel use this same
order, so that depths, radii and bounding boxes are bit-identical between the three and with the
reference (tests/test_oracle_golden.py: 0 differing bits at N = 1e5 and 1e6).
"""
from __future__ import annotations

import math
from typing import Dict, NamedTuple, Optional

import numpy as np

f32 = np.float32

# Constants the reference hard-codes ("Config / flags").
MIN_Z = f32(0.2)            # splat/utils.py:294
FOV_CLAMP = f32(1.3)        # splat/utils.py:336-337
DET_FLOOR = f32(1e-3)       # splat/utils.py:387
EIG_FLOOR = f32(0.1)        # splat/utils.py:414
SIGMA_EXTENT = f32(3.0)     # splat/utils.py:421
STOP_T = f32(0.000001)      # splat/gaussian_scene.py:153
ZNEAR = f32(0.001)          # splat/image.py:47
ZFAR = f32(100.0)           # splat/image.py:46


class Camera(NamedTuple):
    """Already-computed float32 camera constants (what ``GsxCamera`` carries)."""

    world2view: np.ndarray   # (4,4) row-vector form, splat/image.py:51-53
    full_proj: np.ndarray    # (4,4) splat/image.py:61-65
    tan_fovx: np.float32     # splat/image.py:42
    tan_fovy: np.float32     # splat/image.py:43
    fx: np.float32           # splat/image.py:28
    fy: np.float32           # splat/image.py:29
    width: int               # splat/image.py:38
    height: int              # splat/image.py:37


class Preprocessed(NamedTuple):
    """Depth-sorted stage-1 output; field names follow splat/schema.py:13-25."""

    points: np.ndarray                  # (Nv,2) == points_xy
    colors: np.ndarray                  # (Nv,3)
    covariance_2d: np.ndarray           # (Nv,2,2)
    depths: np.ndarray                  # (Nv,)
    inverse_covariance_2d: np.ndarray   # (Nv,2,2)
    radius: np.ndarray                  # (Nv,)
    points_xy: np.ndarray               # (Nv,2)
    min_x: np.ndarray
    min_y: np.ndarray
    max_x: np.ndarray
    max_y: np.ndarray
    sigmoid_opacity: np.ndarray         # (Nv,1)
    order: np.ndarray                   # (Nv,) original Gaussian index of each sorted row


def fma(a, b, c) -> np.ndarray:
    """float32 fused multiply-add, correctly rounded: the product of two float32 is exact in float64; the
    float64 sum is rounded to odd (TwoSum tells which way the exact sum lies), which makes the final
    rounding to float32 immune to double rounding (53 >= 2 * 24 + 2)."""
    p = np.asarray(a, np.float32).astype(np.float64) * np.asarray(b, np.float32).astype(np.float64)
    c = np.broadcast_to(np.asarray(c, np.float32).astype(np.float64), p.shape)
    s = p + c
    t = s - p
    e = (p - (s - t)) + (c - t)
    bits = np.array(s, dtype=np.float64, copy=True).view(np.int64)
    fix = (e != 0.0) & ((bits & 1) == 0) & np.isfinite(s)
    step = np.where((e > 0.0) == (s > 0.0), 1, -1)
    bits += np.where(fix, step, 0)
    return bits.view(np.float64).astype(np.float32)


# --------------------------------------------------------------------------- camera

def rotation_from_quaternion(q: np.ndarray) -> np.ndarray:
    """(N,4) (w,x,y,z) -> (N,3,3).  splat/utils.py:132-155 (normalises first)."""
    q = np.asarray(q, dtype=f32)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    norm = np.sqrt(w * w + x * x + y * y + z * z)
    w, x, y, z = w / norm, x / norm, y / norm, z / norm
    one, two = f32(1.0), f32(2.0)
    R = np.empty((q.shape[0], 3, 3), dtype=f32)
    R[:, 0, 0] = one - two * (y * y + z * z)
    R[:, 0, 1] = two * (x * y - w * z)
    R[:, 0, 2] = two * (x * z + w * y)
    R[:, 1, 0] = two * (x * y + w * z)
    R[:, 1, 1] = one - two * (x * x + z * z)
    R[:, 1, 2] = two * (y * z - w * x)
    R[:, 2, 0] = two * (x * z - w * y)
    R[:, 2, 1] = two * (y * z + w * x)
    R[:, 2, 2] = one - two * (x * x + y * y)
    return R


def build_camera(qvec, tvec, fx, fy, width: int, height: int) -> Camera:
    """COLMAP PINHOLE pose/intrinsics -> camera constants.

    splat/image.py:28-66 with splat/utils.py:158-159 (focal2fov), :162-172
    (getWorld2View), :189-225 (getProjectionMatrix).  The principal point is ignored by
    the render path.  ``math.atan``/``math.tan`` run in double on float32 operands exactly
    as the reference does.
    """
    fx32, fy32 = f32(fx), f32(fy)
    w32, h32 = f32(width), f32(height)
    R = rotation_from_quaternion(np.asarray(qvec, dtype=f32)[None, :])[0]
    E = np.zeros((4, 4), dtype=f32)
    E[:3, :3] = R
    E[:3, 3] = np.asarray(tvec, dtype=f32)
    E[3, 3] = f32(1.0)
    V = np.ascontiguousarray(E.T)

    fovx = f32(2.0 * math.atan(float(w32 / (f32(2.0) * fx32))))
    fovy = f32(2.0 * math.atan(float(h32 / (f32(2.0) * fy32))))
    tanx = np.tan(fovx / f32(2.0), dtype=f32)
    tany = np.tan(fovy / f32(2.0), dtype=f32)

    thx = f32(math.tan(float(fovx / f32(2.0))))
    thy = f32(math.tan(float(fovy / f32(2.0))))
    top = thy * ZNEAR
    bottom = -top
    right = thx * ZNEAR
    left = -right
    P = np.zeros((4, 4), dtype=f32)
    P[0, 0] = f32(2.0) * ZNEAR / (right - left)
    P[1, 1] = f32(2.0) * ZNEAR / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = f32(1.0)
    P[2, 2] = f32(1.0) * ZFAR / (ZFAR - ZNEAR)
    P[2, 3] = -(ZFAR * ZNEAR) / (ZFAR - ZNEAR)
    Pt = np.ascontiguousarray(P.T)
    F = np.zeros((4, 4), dtype=f32)
    for i in range(4):
        for j in range(4):
            acc = V[i, 0] * Pt[0, j]
            for k in range(1, 4):
                acc = f32(acc + V[i, k] * Pt[k, j])
            F[i, j] = acc
    return Camera(V, F, f32(tanx), f32(tany), fx32, fy32, int(width), int(height))


# --------------------------------------------------------------------------- stage 1

def _row4(p: np.ndarray, M: np.ndarray, col: int, view_rows: Optional[int] = None) -> np.ndarray:
    """Column ``col`` of ``[p,1] @ M`` as torch's (N,4) @ (4,4) executes it: a sequential FMA chain
    (splat/gaussian_scene.py:79-90, splat/utils.py:305-307, 333).  ``view_rows``: M is ``world2view`` -- a TRANSPOSED
    view in the reference (splat/image.py:51-53) -- and the product has that many rows: with one row MKL computes
    ((p0 M0 + p1 M1 fused) + M3) + p2 M2, with two or three (p0 M0 + p2 M2) + (p1 M1 + M3) unfused
    (oracle/probe_torch_order.py); ``full_proj_transform`` is contiguous and always takes the chain."""
    if view_rows is not None and view_rows == 1:
        return (fma(p[:, 1], M[1, col], p[:, 0] * M[0, col]) + M[3, col]) + p[:, 2] * M[2, col]
    if view_rows is not None and view_rows <= 3:
        return (p[:, 0] * M[0, col] + p[:, 2] * M[2, col]) + (p[:, 1] * M[1, col] + M[3, col])
    acc = p[:, 0] * M[0, col]
    acc = fma(p[:, 1], M[1, col], acc)
    acc = fma(p[:, 2], M[2, col], acc)
    return acc + M[3, col]


def covariance_3d(scales: np.ndarray, quats: np.ndarray) -> np.ndarray:
    """Sigma = (R S)(R S)^T.  splat/gaussians.py:54-69 (F.normalize then build_rotation)."""
    q = np.asarray(quats, dtype=f32)
    n1 = np.sqrt(((q[:, 0] * q[:, 0] + q[:, 1] * q[:, 1]) + q[:, 2] * q[:, 2]) + q[:, 3] * q[:, 3])
    n1 = np.maximum(n1, f32(1e-12))
    R = rotation_from_quaternion(q / n1[:, None])
    s = np.asarray(scales, dtype=f32)
    M = R * s[:, None, :]
    S = np.empty_like(M)
    for i in range(3):
        for j in range(3):
            S[:, i, j] = (M[:, i, 0] * M[:, j, 0] + M[:, i, 1] * M[:, j, 1]) + M[:, i, 2] * M[:, j, 2]
    return S


def _mm3(A: np.ndarray, B: np.ndarray) -> np.ndarray:
    """Batched (N,3,3) @ (N,3,3): each entry accumulated k=0,1,2 left to right, every product and sum rounded."""
    C = np.empty_like(A)
    for i in range(3):
        for j in range(3):
            C[:, i, j] = (A[:, i, 0] * B[:, 0, j] + A[:, i, 1] * B[:, 1, j]) + A[:, i, 2] * B[:, 2, j]
    return C


def _mm3_single(A: np.ndarray, B: np.ndarray, small: bool = False) -> np.ndarray:
    """(N,3,3) @ one (3,3): torch folds it into a (3N,3) @ (3,3) sgemm, a sequential FMA chain over k.
    ``small``: ``X @ W.T`` (W.T = world2view[:3,:3], a column-major view) with N <= 3 goes to another MKL kernel,
    (k0 + k2) + k1 with nothing fused."""
    C = np.empty_like(A)
    for i in range(3):
        for j in range(3):
            if small:
                C[:, i, j] = (A[:, i, 0] * B[0, j] + A[:, i, 2] * B[2, j]) + A[:, i, 1] * B[1, j]
            else:
                C[:, i, j] = fma(A[:, i, 2], B[2, j], fma(A[:, i, 1], B[1, j], A[:, i, 0] * B[0, j]))
    return C


def covariance_2d(points: np.ndarray, cov3d: np.ndarray, cam: Camera) -> np.ndarray:
    """EWA projection of Sigma.  splat/utils.py:320-354.  The rows given are the batch the reference
    multiplies at once (its N_vis), which selects the kernel of ``... @ W.T`` (see ``_mm3_single``)."""
    V = cam.world2view
    rows = points.shape[0]
    tx, ty, tz = _row4(points, V, 0, rows), _row4(points, V, 1, rows), _row4(points, V, 2, rows)
    limx = FOV_CLAMP * cam.tan_fovx
    limy = FOV_CLAMP * cam.tan_fovy
    x = np.minimum(np.maximum(tx / tz, -limx), limx) * tz
    y = np.minimum(np.maximum(ty / tz, -limy), limy) * tz
    n = points.shape[0]
    J = np.zeros((n, 3, 3), dtype=f32)
    J[:, 0, 0] = cam.fx / tz
    J[:, 0, 2] = -(cam.fx * x) / (tz * tz)
    J[:, 1, 1] = cam.fy / tz
    J[:, 1, 2] = -(cam.fy * y) / (tz * tz)
    Wm = np.ascontiguousarray(V[:3, :3].T)
    A = _mm3_single(J, Wm)
    B = _mm3(A, cov3d)
    C = _mm3_single(B, np.ascontiguousarray(Wm.T), small=n <= 3)
    D = _mm3(C, np.ascontiguousarray(np.transpose(J, (0, 2, 1))))
    return np.ascontiguousarray(D[:, :2, :2])


def inverted_covariance(c2: np.ndarray) -> np.ndarray:
    """splat/utils.py:368-393 (determinant floored at 1e-3; entries divided one by one)."""
    det = c2[:, 0, 0] * c2[:, 1, 1] - c2[:, 0, 1] * c2[:, 1, 0]
    det = np.maximum(det, DET_FLOOR)
    inv = np.empty_like(c2)
    inv[:, 0, 0] = c2[:, 1, 1] / det
    inv[:, 1, 1] = c2[:, 0, 0] / det
    inv[:, 0, 1] = -c2[:, 0, 1] / det
    inv[:, 1, 0] = -c2[:, 1, 0] / det
    return inv


def extent_radius(c2: np.ndarray) -> np.ndarray:
    """splat/utils.py:409-423: r = ceil(3 sqrt(lambda_max)), discriminant floored at 0.1."""
    mid = f32(0.5) * (c2[:, 0, 0] + c2[:, 1, 1])
    det = c2[:, 0, 0] * c2[:, 1, 1] - c2[:, 0, 1] * c2[:, 0, 1]
    m = np.maximum(mid * mid - det, EIG_FLOOR)
    root = np.sqrt(m)
    lam = np.maximum(mid + root, mid - root)
    return np.ceil(SIGMA_EXTENT * np.sqrt(lam))


def sigmoid(x: np.ndarray) -> np.ndarray:
    """``torch.sigmoid`` on ONE element at a time (splat/gaussian_scene.py:164: the reference's second sigmoid) -- torch's
    scalar path, libm's ``expf``.  Arrays of up to 2^16 elements (a tile's list) go through libm itself; longer ones
    through numpy's float32 exp, which is a last bit off here and there (nothing on the pinned paths is that long)."""
    x = np.asarray(x, dtype=f32)
    if 0 < x.size <= 65536:
        return (f32(1.0) / (f32(1.0) + _libm_expf(f32(0.0) - x).reshape(x.shape))).astype(f32)
    return (f32(1.0) / (f32(1.0) + np.exp(-x))).astype(f32)


def _vexp(d: np.ndarray) -> np.ndarray:
    """The SIMD exponential inside torch's vectorised sigmoid (Sleef's 1.0-ulp expf, FMA throughout)."""
    d = np.asarray(d, f32)
    q = np.rint(d * f32(1.442695040888963407359924681001892137426645954152985934135449406931)).astype(np.int32)
    qf = q.astype(f32)
    s = fma(qf, f32(-0.693145751953125), d)
    s = fma(qf, f32(-1.428606765330187045e-06), s)
    u = np.full_like(s, f32(0.000198527617612853646278381))
    for c in (0.00139304355252534151077271, 0.00833336077630519866943359, 0.0416664853692054748535156,
              0.166666671633720397949219, 0.5):
        u = fma(u, s, f32(c))
    u = f32(1.0) + fma(s * s, u, s)
    p2 = lambda e: ((e + 0x7F).astype(np.uint32) << np.uint32(23)).view(f32)  # noqa: E731
    u = u * p2(q >> 1) * p2(q - (q >> 1))
    u = np.where(d < f32(-104.0), f32(0.0), u)
    return np.where(d > f32(100.0), f32(np.inf), u).astype(f32)


_LIBM = None


def _libm_expf(v: np.ndarray) -> np.ndarray:
    """glibc's ``expf`` on every element (what torch's scalar tail loop calls; oracle/raster_cpu.c links the same one)."""
    global _LIBM
    if _LIBM is None:
        import ctypes
        import ctypes.util

        _LIBM = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
        _LIBM.expf.restype = ctypes.c_float
        _LIBM.expf.argtypes = [ctypes.c_float]
    return np.array([_LIBM.expf(float(t)) for t in np.asarray(v, f32).reshape(-1)], dtype=f32)


def sigmoid_torch(x: np.ndarray, threads: int = 8) -> np.ndarray:
    """``torch.sigmoid`` on a contiguous float32 array as torch 2.10 executes it on an AVX-512 host
    (splat/gaussian_scene.py:143; probed: the reference's bits): ``1 / (1 + e)`` with ``e`` the SIMD exponential on
    whole groups of 32 elements and libm's ``expf`` on what is left at the end of every thread's chunk -- chunks of
    ``ceil(n / t)`` elements, ``t = min(threads, ceil(n / 32768))``.  The value therefore depends on the element's
    position and on the thread count of the reference run (8 where the fixtures were made)."""
    x = np.asarray(x, dtype=f32)
    flat = x.reshape(-1)
    n = flat.size
    out = (f32(1.0) / (f32(1.0) + _vexp(f32(0.0) - flat))).astype(f32)
    if n:
        parts = max(1, min(int(threads), -(-n // 32768)))
        chunk = -(-n // parts)
        i = np.arange(n)
        begin = (i // chunk) * chunk
        length = np.minimum(n - begin, chunk)
        tail = (i - begin) >= length - length % 32
        # libm's expf ITSELF, element by element (at most 31 per chunk): the float64 exponential rounded once stood in
        # for it through round 5 -- glibc's expf is within 0.502 ulp, not correctly rounded, and a 200-case run of
        # oracle/fuzz_vs_reference.py (round 6) met two elements in ~50 000 tails where the two differ by a bit;
        # numpy's own float32 exp is another SIMD routine again
        out[tail] = f32(1.0) / (f32(1.0) + _libm_expf(f32(0.0) - flat[tail]))
    return out.reshape(x.shape)


def preprocess(points, colors, scales, quats, opacity_logit, cam: Camera,
               order: Optional[np.ndarray] = None) -> Preprocessed:
    """Stage 1.  splat/gaussian_scene.py:70-144.

    ``colors`` are the stored colours (already divided by 256, splat/gaussians.py:20-22).
    Depth order is ascending view-space z with ties broken by original index (the
    reference's ``torch.argsort`` leaves ties implementation-defined);
    ``order`` overrides the permutation (original indices of in-view Gaussians) so a
    fixture can pin the reference's own permutation.
    """
    points = np.asarray(points, dtype=f32)
    V, F = cam.world2view, cam.full_proj
    zv = _row4(points, V, 2, points.shape[0])               # the cull multiplies all n points at once
    in_view = zv >= MIN_Z                                   # splat/utils.py:293-310
    idx = np.nonzero(in_view)[0]
    p = points[idx]
    cov3 = covariance_3d(np.asarray(scales, f32), np.asarray(quats, f32))[idx]

    depth = _row4(p, V, 2, idx.size)                        # ... everything after it the visible ones
    cw = _row4(p, F, 3)
    ndc_x = _row4(p, F, 0) / cw
    ndc_y = _row4(p, F, 1) / cw
    # splat/utils.py:313-317: (v + 1) * (dim - 1) * 0.5
    x_pix = (ndc_x + f32(1.0)) * (f32(cam.width) - f32(1.0)) * f32(0.5)
    y_pix = (ndc_y + f32(1.0)) * (f32(cam.height) - f32(1.0)) * f32(0.5)
    xy = np.stack([x_pix, y_pix], axis=1).astype(f32)

    c2 = covariance_2d(p, cov3, cam)
    inv = inverted_covariance(c2)
    r = extent_radius(c2)
    min_x, max_x = np.floor(x_pix - r), np.ceil(x_pix + r)
    min_y, max_y = np.floor(y_pix - r), np.ceil(y_pix + r)

    if order is None:
        perm = np.lexsort((idx, depth.view(np.uint32)))     # depth >= 0.2 > 0: bits are monotone
    else:
        lookup = -np.ones(points.shape[0], dtype=np.int64)
        lookup[idx] = np.arange(idx.size)
        perm = lookup[np.asarray(order, dtype=np.int64)]
        assert (perm >= 0).all() and perm.size == idx.size
    cols = np.asarray(colors, dtype=f32)[idx]
    op = np.asarray(opacity_logit, dtype=f32).reshape(-1, 1)[idx]
    return Preprocessed(
        points=xy[perm], colors=cols[perm], covariance_2d=c2[perm], depths=depth[perm],
        inverse_covariance_2d=inv[perm], radius=r[perm], points_xy=xy[perm],
        min_x=min_x[perm], min_y=min_y[perm], max_x=max_x[perm], max_y=max_y[perm],
        sigmoid_opacity=sigmoid_torch(op[perm]), order=idx[perm],
    )


# --------------------------------------------------------------------------- stage 2

def tile_origins(extent: int, tile: int):
    """``range(0, extent - tile, tile)``: the last tile row/column is never rendered.
    splat/gaussian_scene.py:208,214."""
    return range(0, extent - tile, tile)


def tile_list(pre, x0: int, y0: int, tile: int) -> np.ndarray:
    """Indices (depth order) binned into the tile at (x0,y0).  splat/gaussian_scene.py:209-218."""
    m = (pre.min_x <= f32(x0 + tile)) & (pre.max_x >= f32(x0)) & \
        (pre.min_y <= f32(y0 + tile)) & (pre.max_y >= f32(y0))
    return np.nonzero(m)[0]


def render_pixel_scalar(px: int, py: int, means, colors, sig_op, inv) -> np.ndarray:
    """One pixel, scalar loop.  splat/gaussian_scene.py:146-171 + splat/utils.py:357-365.

    ``sig_op`` is already sigmoid(opacity); the reference applies sigmoid again (:164).
    """
    T = f32(1.0)
    C = np.zeros(3, dtype=f32)
    fx_, fy_ = f32(px), f32(py)
    half = f32(-0.5)
    for k in range(means.shape[0]):
        d0 = half * (means[k, 0] - fx_)
        d1 = half * (means[k, 1] - fy_)
        # (1,2) @ (2,2) is an sgemm: one FMA per output; the (1,2) @ (2,1) after it rounds both products
        t0 = f32(fma(d1, inv[k, 1, 0], d0 * inv[k, 0, 0]))
        t1 = f32(fma(d1, inv[k, 1, 1], d0 * inv[k, 0, 1]))
        e0 = means[k, 0] - fx_
        e1 = means[k, 1] - fy_
        power = f32(t0 * e0 + t1 * e1)
        w = np.exp(power, dtype=f32)
        o2 = f32(1.0) / (f32(1.0) + _libm_expf(f32(0.0) - sig_op[k:k + 1])[0])
        alpha = f32(w * o2)
        test = f32(T * (f32(1.0) - alpha))
        if test < STOP_T:
            return C
        C = C + f32(T * alpha) * colors[k]
        T = test
    return C


def render_tile_vector(x0: int, y0: int, tile: int, means, colors, sig_op, inv) -> np.ndarray:
    """All tile*tile pixels of one tile at once (same arithmetic as render_pixel_scalar).
    Returns (tile, tile, 3) indexed [x - x0, y - y0].  splat/gaussian_scene.py:173-198."""
    xs = (np.arange(x0, x0 + tile, dtype=np.int64)).astype(f32)
    ys = (np.arange(y0, y0 + tile, dtype=np.int64)).astype(f32)
    PX, PY = np.meshgrid(xs, ys, indexing="ij")
    T = np.ones((tile, tile), dtype=f32)
    C = np.zeros((tile, tile, 3), dtype=f32)
    live = np.ones((tile, tile), dtype=bool)
    half = f32(-0.5)
    o2 = sigmoid(sig_op.reshape(-1))
    for k in range(means.shape[0]):
        e0 = means[k, 0] - PX
        e1 = means[k, 1] - PY
        d0 = half * e0
        d1 = half * e1
        t0 = fma(d1, inv[k, 1, 0], d0 * inv[k, 0, 0])
        t1 = fma(d1, inv[k, 1, 1], d0 * inv[k, 0, 1])
        w = np.exp(t0 * e0 + t1 * e1)
        alpha = w * o2[k]
        test = T * (f32(1.0) - alpha)
        live &= ~(test < STOP_T)
        if not live.any():
            break
        contrib = (T * alpha)[..., None] * colors[k][None, None, :]
        C = np.where(live[..., None], C + contrib, C)
        T = np.where(live, test, T)
    return C


def render_image(pre, width: int, height: int, tile: int = 16, scalar: bool = False,
                 window=None, stats: Optional[Dict] = None) -> np.ndarray:
    """Stage 2.  splat/gaussian_scene.py:200-238.  Returns (W,H,3) float32 indexed [x,y];
    pixels of the never-rendered last tile row/column stay 0.

    ``window=(tx0,tx1,ty0,ty1)`` restricts rendering to a tile-index window (used by the
    bounded CPU-baseline sample in bench.py); ``stats['pairs']`` counts list entries x pixels.
    """
    image = np.zeros((width, height, 3), dtype=f32)
    pairs = 0
    for ix, x0 in enumerate(tile_origins(width, tile)):
        if window is not None and not (window[0] <= ix < window[1]):
            continue
        for iy, y0 in enumerate(tile_origins(height, tile)):
            if window is not None and not (window[2] <= iy < window[3]):
                continue
            sel = tile_list(pre, x0, y0, tile)
            if sel.size == 0:
                continue
            pairs += sel.size * tile * tile
            m, c = pre.points[sel], pre.colors[sel]
            o, q = pre.sigmoid_opacity[sel].reshape(-1), pre.inverse_covariance_2d[sel]
            if scalar:
                for px in range(x0, x0 + tile):
                    for py in range(y0, y0 + tile):
                        image[px, py] = render_pixel_scalar(px, py, m, c, o, q)
            else:
                image[x0:x0 + tile, y0:y0 + tile] = render_tile_vector(x0, y0, tile, m, c, o, q)
    if stats is not None:
        stats["pairs"] = pairs
    return image


# --------------------------------------------------------------------------- SH (build extension)

_SH_C0 = 0.28209479177387814
_SH_C1 = 0.4886025119029199
_SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
_SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
          1.445305721320277, -0.5900435899266435)


def sh_to_rgb(points, sh, degree: int, camera_center) -> np.ndarray:
    """Spherical harmonics -> RGB in the published 3D Gaussian Splatting convention (Kerbl et al.
    2023, ``eval_sh``; not part of /root/reference, which has no SH at all -- PARITY UNPINNED).
    float64 inside, so it is an independent check of the float32 kernel."""
    p = np.asarray(points, np.float64)
    c = np.asarray(sh, np.float64)
    d = p - np.asarray(camera_center, np.float64)[None, :]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    res = _SH_C0 * c[:, 0]
    if degree > 0:
        res = res - _SH_C1 * y * c[:, 1] + _SH_C1 * z * c[:, 2] - _SH_C1 * x * c[:, 3]
    if degree > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + _SH_C2[0] * xy * c[:, 4] + _SH_C2[1] * yz * c[:, 5] + _SH_C2[2] * (2 * zz - xx - yy) * c[:, 6]
               + _SH_C2[3] * xz * c[:, 7] + _SH_C2[4] * (xx - yy) * c[:, 8])
        if degree > 2:
            res = (res + _SH_C3[0] * y * (3 * xx - yy) * c[:, 9] + _SH_C3[1] * xy * z * c[:, 10]
                   + _SH_C3[2] * y * (4 * zz - xx - yy) * c[:, 11]
                   + _SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * c[:, 12]
                   + _SH_C3[4] * x * (4 * zz - xx - yy) * c[:, 13] + _SH_C3[5] * z * (xx - yy) * c[:, 14]
                   + _SH_C3[6] * x * (xx - 3 * yy) * c[:, 15])
    return np.maximum(res + 0.5, 0.0).astype(np.float32)
