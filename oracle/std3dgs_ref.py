"""numpy restatement of the forward pass of the published 3D Gaussian Splatting rasteriser
(``GSX_SEM_STD_3DGS``, a build extension -- SURVEY.md section 8(f) rank 3).

TEST INFRASTRUCTURE ONLY (same rule as the rest of ``oracle/``).

Parity status: **PARITY UNPINNED**.  The algorithm is the one published with "3D Gaussian Splatting
for Real-Time Radiance Field Rendering" (Kerbl, Kopanas, Leimkuehler, Drettakis, SIGGRAPH 2023) and
implemented by its CUDA rasteriser (``diff-gaussian-rasterization``, forward pass).  That code is
neither part of ``/root/reference`` nor a pinned dependency of it (``pyproject.toml`` lists no
rasteriser) and it cannot be built here (CUDA), so there are no golden vectors; this module and
``oracle/raster_cpu.c:orc_render_std3dgs`` restate the published steps independently of each other
(vectorised numpy per tile here, scalar C there) and ``tests/test_std3dgs_oracle.py`` checks them
against each other.

Steps (float32, explicit operation order shared with the C restatement and the HIP stage 1):
  stage 1   cull z_view <= 0.2; q normalised once; Sigma = (R S)(R S)^T; p_w = 1/(w + 1e-7);
            pixel = ((ndc + 1) extent - 1)/2; focal = extent / (2 tan(fov/2)); EWA with the view-space
            point clamped to 1.3 tan(fov/2); cov00 += 0.3, cov11 += 0.3; det == 0 dropped;
            conic = (c, -b, a)/det; lambda = mid +- sqrt(max(0.1, mid^2 - det));
            r = ceil(3 sqrt(lambda_max)); tile rectangle [(int)((p - r)/T), (int)((p + r + T - 1)/T))
            clamped to the grid, empty rectangles dropped; opacity = sigmoid(logit).
  order     per tile by view depth, ties by Gaussian index.
  stage 2   per pixel (integer coordinates): power = -0.5 (A dx^2 + C dy^2) - B dx dy; skipped when
            power > 0; alpha = min(0.99, opacity exp(power)); skipped when alpha < 1/255; the pixel
            stops (before accumulating) when T (1 - alpha) < 1e-4; C += c alpha T; out = C + T bg.
"""
from __future__ import annotations

from typing import NamedTuple, Optional, Tuple

import numpy as np

from .cpu_ref import Camera, _mm3, _mm3_single, _row4, sigmoid

f32 = np.float32


class Stage1(NamedTuple):
    xy: np.ndarray        # (n,2) pixel position
    conic: np.ndarray     # (n,3) A, B, C
    radius: np.ndarray    # (n,)
    depth: np.ndarray     # (n,)
    opacity: np.ndarray   # (n,)
    rect: np.ndarray      # (n,4) int tile rectangle lx, hx (exclusive), ly, hy (exclusive)
    keep: np.ndarray      # (n,) bool: in front of the camera, det != 0, rectangle not empty
    in_front: np.ndarray  # (n,) bool: z_view > 0.2


def _rotation(q: np.ndarray) -> np.ndarray:
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
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


def _tile_index(v: np.ndarray, nt: int) -> np.ndarray:
    """min(nt, max(0, (int)v)) with (int) truncating toward zero; -1 for NaN (no tile)."""
    nan = np.isnan(v)
    c = np.trunc(np.clip(np.where(nan, f32(0.0), v), f32(-1073741824.0), f32(1073741824.0))).astype(np.int64)
    return np.where(nan, -1, np.clip(c, 0, nt))


def stage1(points, scales, quats, opacity_logit, cam: Camera, tile: int = 16) -> Stage1:
    p = np.ascontiguousarray(np.asarray(points, f32).reshape(-1, 3))
    s = np.asarray(scales, f32).reshape(-1, 3)
    q = np.asarray(quats, f32).reshape(-1, 4)
    n = p.shape[0]
    V, F = cam.world2view, cam.full_proj
    W, H = int(cam.width), int(cam.height)
    with np.errstate(all="ignore"):
        tz = _row4(p, V, 2)
        in_front = tz > f32(0.2)
        n1 = np.sqrt(((q[:, 0] * q[:, 0] + q[:, 1] * q[:, 1]) + q[:, 2] * q[:, 2]) + q[:, 3] * q[:, 3])
        n1 = np.maximum(n1, f32(1e-12))
        R = _rotation(q / n1[:, None])
        M = R * s[:, None, :]
        S = np.empty_like(M)
        for i in range(3):
            for j in range(3):
                S[:, i, j] = (M[:, i, 0] * M[:, j, 0] + M[:, i, 1] * M[:, j, 1]) + M[:, i, 2] * M[:, j, 2]
        pw = f32(1.0) / (_row4(p, F, 3) + f32(0.0000001))
        ndcx, ndcy = _row4(p, F, 0) * pw, _row4(p, F, 1) * pw
        x = ((ndcx + f32(1.0)) * f32(W) - f32(1.0)) * f32(0.5)
        y = ((ndcy + f32(1.0)) * f32(H) - f32(1.0)) * f32(0.5)
        fx = f32(W) / (f32(2.0) * f32(cam.tan_fovx))
        fy = f32(H) / (f32(2.0) * f32(cam.tan_fovy))
        tx, ty = _row4(p, V, 0), _row4(p, V, 1)
        limx, limy = f32(1.3) * f32(cam.tan_fovx), f32(1.3) * f32(cam.tan_fovy)
        cx = np.minimum(np.maximum(tx / tz, -limx), limx) * tz
        cy = np.minimum(np.maximum(ty / tz, -limy), limy) * tz
        J = np.zeros((n, 3, 3), dtype=f32)
        J[:, 0, 0] = fx / tz
        J[:, 0, 2] = -(fx * cx) / (tz * tz)
        J[:, 1, 1] = fy / tz
        J[:, 1, 2] = -(fy * cy) / (tz * tz)
        Wm = np.ascontiguousarray(V[:3, :3].T)
        # the same operation order as the reference-rule projection (cpu_ref.covariance_2d): one-matrix factors fused
        D = _mm3(_mm3_single(_mm3(_mm3_single(J, Wm), S), np.ascontiguousarray(Wm.T)), np.ascontiguousarray(np.transpose(J, (0, 2, 1))))
        ca, cb, cd = D[:, 0, 0] + f32(0.3), D[:, 0, 1], D[:, 1, 1] + f32(0.3)
        det = ca * cd - cb * cb
        det_inv = f32(1.0) / det
        conic = np.stack([cd * det_inv, -cb * det_inv, ca * det_inv], axis=1).astype(f32)
        mid = f32(0.5) * (ca + cd)
        root = np.sqrt(np.fmax(f32(0.1), mid * mid - det))
        lam = np.fmax(mid + root, mid - root)
        r = np.ceil(f32(3.0) * np.sqrt(lam))
        T = f32(tile)
        ntx, nty = (W + tile - 1) // tile, (H + tile - 1) // tile
        lx = _tile_index((x - r) / T, ntx)
        hx = _tile_index(((x + r + T) - f32(1.0)) / T, ntx)
        ly = _tile_index((y - r) / T, nty)
        hy = _tile_index(((y + r + T) - f32(1.0)) / T, nty)
        valid = (lx >= 0) & (hx >= 0) & (ly >= 0) & (hy >= 0)
        area = np.where(valid, (hx - lx) * (hy - ly), 0)
        keep = in_front & (det != 0) & (area > 0)
        op = sigmoid(np.asarray(opacity_logit, f32).reshape(-1))
    return Stage1(xy=np.stack([x, y], axis=1).astype(f32), conic=conic, radius=r.astype(f32), depth=tz.astype(f32),
                  opacity=op.astype(f32), rect=np.stack([lx, hx, ly, hy], axis=1), keep=keep, in_front=in_front)


def render(points, colors, scales, quats, opacity_logit, cam: Camera, tile: int = 16,
           background=(0.0, 0.0, 0.0), window: Optional[Tuple[int, int, int, int]] = None):
    """Returns (image (H,W,3) indexed [y,x], n_visible, instances)."""
    st = stage1(points, scales, quats, opacity_logit, cam, tile)
    col = np.asarray(colors, f32).reshape(-1, 3)
    W, H = int(cam.width), int(cam.height)
    ntx, nty = (W + tile - 1) // tile, (H + tile - 1) // tile
    bg = np.asarray(background, f32)
    image = np.zeros((H, W, 3), f32)
    idx = np.nonzero(st.keep)[0]
    order = idx[np.argsort(st.depth[idx], kind="stable")]          # depth, ties by Gaussian index
    rect = st.rect[order]
    wx0, wx1, wy0, wy1 = (0, ntx, 0, nty) if window is None else window
    wx1 = ntx if wx1 <= 0 or wx1 > ntx else wx1
    wy1 = nty if wy1 <= 0 or wy1 > nty else wy1
    instances = 0
    for tix in range(max(wx0, 0), wx1):
        in_x = (rect[:, 0] <= tix) & (tix < rect[:, 1])
        for tiy in range(max(wy0, 0), wy1):
            lst = order[in_x & (rect[:, 2] <= tiy) & (tiy < rect[:, 3])]
            instances += len(lst)
            x0, y0 = tix * tile, tiy * tile
            xs = np.arange(x0, min(x0 + tile, W), dtype=f32)
            ys = np.arange(y0, min(y0 + tile, H), dtype=f32)
            PX, PY = np.meshgrid(xs, ys)                           # (h, w)
            T = np.ones(PX.shape, f32)
            C = np.zeros(PX.shape + (3,), f32)
            done = np.zeros(PX.shape, bool)
            for g in lst:
                if done.all():
                    break
                A, B, Cc = st.conic[g]
                dx, dy = st.xy[g, 0] - PX, st.xy[g, 1] - PY
                power = f32(-0.5) * (A * dx * dx + Cc * dy * dy) - B * dx * dy
                with np.errstate(over="ignore", under="ignore", invalid="ignore"):
                    alpha = np.minimum(f32(0.99), st.opacity[g] * np.exp(power, dtype=f32)).astype(f32)
                    use = ~(power > 0) & ~(alpha < f32(1.0 / 255.0)) & ~done
                    test = (T * (f32(1.0) - alpha)).astype(f32)
                stop = use & (test < f32(0.0001))
                done |= stop
                acc = use & ~stop
                for ch in range(3):
                    C[..., ch] = np.where(acc, C[..., ch] + col[g, ch] * alpha * T, C[..., ch])
                T = np.where(acc, test, T)
            image[y0:y0 + len(ys), x0:x0 + len(xs)] = C + T[..., None] * bg
    return image, int(st.in_front.sum()), instances
