"""Deterministic synthetic scenes (SURVEY.md section 8(d) generator).

The reference ships no data (``get_data.sh`` needs the network) and never renders more than
52 363 isotropic points, so the benchmark configs of BASELINE.json (1e5 / 1e6 / 5e6 Gaussians)
use this generator: a frozen numpy ``RandomState`` stream, Gaussians placed inside the view
frustum of a COLMAP PINHOLE camera posed like Treehill image 100 (``part_1.ipynb:179``).
"""
from __future__ import annotations

from typing import Dict

import numpy as np

# Treehill image 100 pose, as printed in the reference notebook part_1.ipynb:179.
TREEHILL_QVEC = (0.96282662, -0.23562335, 0.12748722, 0.0345476)
TREEHILL_TVEC = (0.0530637, 0.87330016, 3.58750122)


def _rotation(qvec) -> np.ndarray:
    w, x, y, z = (np.asarray(qvec, dtype=np.float64) / np.linalg.norm(qvec))
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
    ])


def make_scene(n: int, width: int, height: int, seed: int = 0, behind_fraction: float = 0.0,
               qvec=TREEHILL_QVEC, tvec=TREEHILL_TVEC, spread: float = 1.0,
               sigma_scale: float = 1.0, cluster_fraction: float = 0.0, cluster_area: float = 0.05,
               cluster_center=(0.35, -0.2), sigma_ln: float = 0.5) -> Dict[str, np.ndarray]:
    """Returns float32 arrays ``points (n,3)``, ``colors_0_255 (n,3)``, ``scales (n,3)`` (linear),
    ``quaternions (n,4)`` (w,x,y,z, unnormalised), ``opacity (n,1)`` (logit) plus the camera
    ``qvec, tvec, fx, fy, cx, cy, width, height``.

    Draw order: z, u, v, sigma(n,3), q(n,4), opacity(n,1), rgb(n,3).  ``behind_fraction`` > 0
    additionally moves that share of the points behind the z >= 0.2 cull plane (drawn last, so
    the default stream is unchanged).  ``spread`` > 1 places points up to that multiple of the
    frustum's half-width off axis (beyond 1.3 the EWA clamp of splat/utils.py:336-337 is active),
    ``sigma_scale`` scales the footprints; both leave the random stream untouched.

    Heavy-tailed variant (not a BASELINE config; trained scenes are never uniform): ``cluster_fraction`` of
    the Gaussians (chosen by one more uniform draw, after everything else) are moved into a window of
    ``cluster_area`` of the frame around ``cluster_center`` (in [-1,1]^2 frame coordinates), and ``sigma_ln``
    stretches the log-normal footprint distribution from its default 0.5 (same draws, exponent rescaled) --
    ``cluster_fraction=0.5, cluster_area=0.05, sigma_ln=1.0`` gives tile lists whose longest is ~20x the mean.
    """
    rs = np.random.RandomState(seed)
    fx = fy = 0.75 * width
    tanx, tany = width / (2 * fx), height / (2 * fy)
    z = rs.uniform(2.0, 10.0, n)
    u = rs.uniform(-1.0, 1.0, n)
    v = rs.uniform(-1.0, 1.0, n)
    sigma_px = rs.lognormal(np.log(1.5), 0.5, (n, 3))
    quats = rs.normal(0.0, 1.0, (n, 4))
    opacity = rs.normal(0.0, 2.0, (n, 1))
    rgb = rs.uniform(0.0, 255.0, (n, 3))
    if behind_fraction > 0.0:
        behind = rs.uniform(0.0, 1.0, n) < behind_fraction
        z = np.where(behind, -z * 0.5 + 0.15, z)
    if sigma_ln != 0.5:     # same normal draws, another log-standard-deviation
        sigma_px = np.exp(np.log(1.5) + (np.log(sigma_px) - np.log(1.5)) * (sigma_ln / 0.5))
    if cluster_fraction > 0.0:
        inside = rs.uniform(0.0, 1.0, n) < cluster_fraction
        half = np.sqrt(cluster_area)        # the window's half-extent in [-1,1] frame coordinates
        u = np.where(inside, cluster_center[0] + half * u, u)
        v = np.where(inside, cluster_center[1] + half * v, v)
    u, v, sigma_px = u * spread, v * spread, sigma_px * sigma_scale
    p_cam = np.stack([u * tanx * np.abs(z), v * tany * np.abs(z), z], axis=1)
    R, t = _rotation(qvec), np.asarray(tvec, dtype=np.float64)
    # camera = R @ world + t  =>  world = R^T (camera - t)
    world = (p_cam - t[None, :]) @ R
    scales = sigma_px * np.abs(z)[:, None] / fx
    f = np.float32
    return dict(
        points=world.astype(f), colors_0_255=rgb.astype(f), scales=scales.astype(f),
        quaternions=quats.astype(f), opacity=opacity.astype(f),
        qvec=np.asarray(qvec, dtype=np.float64), tvec=np.asarray(tvec, dtype=np.float64),
        fx=np.float64(fx), fy=np.float64(fy), cx=np.float64(width / 2), cy=np.float64(height / 2),
        width=np.int64(width), height=np.int64(height),
    )


def make_trained_like_scene(n: int, width: int, height: int, seed: int = 0, qvec=TREEHILL_QVEC, tvec=TREEHILL_TVEC,
                            sh_degree: int = 3) -> Dict[str, np.ndarray]:
    """A scene with the statistics of a TRAINED 3DGS checkpoint rather than of a uniform cloud (BASELINE config 3 as
    written is a trained Treehill ``.ply``, which is not available offline; not a BASELINE config itself):
      * view-clustered: 65 % of the Gaussians sit in 24 clusters (surfaces) of different size and depth, the rest is
        spread over the frustum (floaters and background);
      * needle and disc footprints: a log-normal size (sigma_ln = 1.2 around 1.2 px) times per-axis factors
        1 / ratio^u, the ratio log-uniform in [1, 50]: axis ratios up to 50:1, random orientation;
      * bimodal opacity: half nearly transparent (logit ~ N(-3, 1)), half nearly opaque (logit ~ N(4, 1.5));
      * degree-``sh_degree`` spherical harmonics (published 3DGS convention): DC from a base colour, higher bands
        ~ N(0, 0.15 / (1 + band)) -- view-dependent colour, clamped at 0 by the evaluation like a trained one.
    Same fields as ``make_scene`` plus ``sh (n, (sh_degree+1)^2, 3)``; a stream of its own (seed + 7919)."""
    rs = np.random.RandomState(seed + 7919)
    fx = fy = 0.75 * width
    tanx, tany = width / (2 * fx), height / (2 * fy)
    n_cl = 24
    cl_u, cl_v = rs.uniform(-0.85, 0.85, n_cl), rs.uniform(-0.85, 0.85, n_cl)
    cl_z, cl_r = rs.uniform(2.5, 9.0, n_cl), np.exp(rs.uniform(np.log(0.02), np.log(0.25), n_cl))
    which = rs.randint(0, n_cl, n)
    clustered = rs.uniform(0.0, 1.0, n) < 0.65
    du, dv, dz = rs.normal(0.0, 1.0, n), rs.normal(0.0, 1.0, n), rs.normal(0.0, 1.0, n)
    u = np.where(clustered, cl_u[which] + cl_r[which] * du, rs.uniform(-1.0, 1.0, n))
    v = np.where(clustered, cl_v[which] + cl_r[which] * dv, rs.uniform(-1.0, 1.0, n))
    z = np.where(clustered, np.maximum(cl_z[which] * (1.0 + 0.05 * dz), 0.5), rs.uniform(2.0, 10.0, n))
    size = rs.lognormal(np.log(1.2), 1.2, n)
    ratio = np.exp(rs.uniform(0.0, np.log(50.0), n))
    expo = rs.uniform(0.0, 1.0, (n, 3))
    expo[np.arange(n), rs.randint(0, 3, n)] = 0.0                     # one axis keeps the full size
    sigma_px = size[:, None] / ratio[:, None] ** expo
    quats = rs.normal(0.0, 1.0, (n, 4))
    faint = rs.uniform(0.0, 1.0, n) < 0.5
    opacity = np.where(faint, rs.normal(-3.0, 1.0, n), rs.normal(4.0, 1.5, n))[:, None]
    rgb = rs.uniform(0.0, 255.0, (n, 3))
    k = (sh_degree + 1) ** 2
    sh = np.zeros((n, k, 3))
    sh[:, 0, :] = (rgb / 256.0 - 0.5) / 0.28209479177387814           # DC: colour = 0.5 + C0 * sh0
    for band in range(1, sh_degree + 1):
        sh[:, band * band:(band + 1) * (band + 1), :] = rs.normal(0.0, 0.15 / (1 + band), (n, 2 * band + 1, 3))
    p_cam = np.stack([u * tanx * z, v * tany * z, z], axis=1)
    R, t = _rotation(qvec), np.asarray(tvec, dtype=np.float64)
    world = (p_cam - t[None, :]) @ R
    scales = sigma_px * z[:, None] / fx
    f = np.float32
    return dict(
        points=world.astype(f), colors_0_255=rgb.astype(f), scales=scales.astype(f), quaternions=quats.astype(f),
        opacity=opacity.astype(f), sh=sh.astype(f), sh_degree=np.int64(sh_degree),
        qvec=np.asarray(qvec, dtype=np.float64), tvec=np.asarray(tvec, dtype=np.float64),
        fx=np.float64(fx), fy=np.float64(fy), cx=np.float64(width / 2), cy=np.float64(height / 2),
        width=np.int64(width), height=np.int64(height),
    )


def make_needle_scene(n: int, width: int, height: int, seed: int = 0, qvec=TREEHILL_QVEC, tvec=TREEHILL_TVEC,
                      long_px=(23.0, 47.0), short_px=(0.1, 0.17)) -> Dict[str, np.ndarray]:
    """Needle footprints (not a BASELINE config): every Gaussian has one axis of ``long_px`` pixels standard deviation
    (3 sigma = 70 .. 140 px) and two of ``short_px`` (3 sigma = 0.3 .. 0.5 px), randomly oriented -- axis ratios of ~250:1, the regime in which the
    three products of ``d Q d^T`` are ~1e4 each and cancel to a few units, so that the float32 operation order of
    the weight (splat/utils.py:363-364) decides the fourth digit of alpha.  Same fields as ``make_scene``; a stream
    of its own (seed + 104729)."""
    rs = np.random.RandomState(seed + 104729)
    fx = fy = 0.75 * width
    tanx, tany = width / (2 * fx), height / (2 * fy)
    z = rs.uniform(2.0, 10.0, n)
    u, v = rs.uniform(-1.0, 1.0, n), rs.uniform(-1.0, 1.0, n)
    sigma_px = np.stack([rs.uniform(long_px[0], long_px[1], n), rs.uniform(short_px[0], short_px[1], n),
                         rs.uniform(short_px[0], short_px[1], n)], axis=1)
    quats = rs.normal(0.0, 1.0, (n, 4))
    opacity = rs.normal(0.0, 2.0, (n, 1))
    rgb = rs.uniform(0.0, 255.0, (n, 3))
    p_cam = np.stack([u * tanx * z, v * tany * z, z], axis=1)
    R, t = _rotation(qvec), np.asarray(tvec, dtype=np.float64)
    world = (p_cam - t[None, :]) @ R
    scales = sigma_px * z[:, None] / fx
    f = np.float32
    return dict(
        points=world.astype(f), colors_0_255=rgb.astype(f), scales=scales.astype(f), quaternions=quats.astype(f),
        opacity=opacity.astype(f), qvec=np.asarray(qvec, dtype=np.float64), tvec=np.asarray(tvec, dtype=np.float64),
        fx=np.float64(fx), fy=np.float64(fy), cx=np.float64(width / 2), cy=np.float64(height / 2),
        width=np.int64(width), height=np.int64(height),
    )


def make_tie_scene(n: int, width: int, height: int, seed: int = 0, levels: int = 33) -> Dict[str, np.ndarray]:
    """Exact view-depth ties (not a BASELINE config): ``make_scene`` under the identity pose -- view depth is then the
    world z itself, bit for bit -- with z snapped to ``levels`` values between 2 and 10 (multiples of 0.25), so that
    every depth is shared by ~n / levels Gaussians that overlap on screen.  What the reference's unstable
    ``torch.argsort`` (splat/gaussian_scene.py:117) does with equal keys is implementation-defined; this scene makes
    its effect on the image visible."""
    sc = make_scene(n=n, width=width, height=height, seed=seed, qvec=(1.0, 0.0, 0.0, 0.0), tvec=(0.0, 0.0, 0.0))
    step = 8.0 / (levels - 1)
    pts = sc["points"].copy()
    pts[:, 2] = (2.0 + np.round((pts[:, 2].astype(np.float64) - 2.0) / step) * step).astype(np.float32)
    sc["points"] = pts
    return sc


def make_few_visible_scene(n: int, width: int, height: int, seed: int = 0, visible: int = 2) -> Dict[str, np.ndarray]:
    """``make_scene`` with all but the first ``visible`` Gaussians moved behind the camera (view depth -1 .. -3): the
    reference then multiplies at most three Jacobians at once in ``J @ W`` (splat/utils.py:354), which its BLAS sums in
    another order than larger batches (oracle/probe_torch_order.py; GSX_FLAG_SMALL_BATCH in include/gsx.h)."""
    sc = make_scene(n=n, width=width, height=height, seed=seed)
    R, t = _rotation(sc["qvec"]), np.asarray(sc["tvec"], dtype=np.float64)
    rs = np.random.RandomState(seed + 15485863)
    m = n - visible
    p_cam = np.stack([rs.uniform(-1.0, 1.0, m), rs.uniform(-1.0, 1.0, m), rs.uniform(-3.0, -1.0, m)], axis=1)
    pts = sc["points"].copy()
    pts[visible:] = ((p_cam - t[None, :]) @ R).astype(np.float32)
    sc["points"] = pts
    return sc


def _qvec_from_rotation(R: np.ndarray) -> np.ndarray:
    """(w, x, y, z) of a rotation matrix (w >= 0), the inverse of ``_rotation``."""
    K = np.array([
        [R[0, 0] - R[1, 1] - R[2, 2], 0.0, 0.0, 0.0],
        [R[1, 0] + R[0, 1], R[1, 1] - R[0, 0] - R[2, 2], 0.0, 0.0],
        [R[2, 0] + R[0, 2], R[2, 1] + R[1, 2], R[2, 2] - R[0, 0] - R[1, 1], 0.0],
        [R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1], R[0, 0] + R[1, 1] + R[2, 2]]]) / 3.0
    vals, vecs = np.linalg.eigh(K)
    q = vecs[[3, 0, 1, 2], int(np.argmax(vals))]
    return -q if q[0] < 0 else q


def orbit_poses(n: int, step_deg: float = 1.0, pivot_depth: float = 6.0, qvec=TREEHILL_QVEC, tvec=TREEHILL_TVEC):
    """``n`` camera poses (qvec, tvec) on an orbit around the point ``pivot_depth`` in front of the base camera --
    the middle of the synthetic scenes' depth range --, ``step_deg`` apart about the camera's vertical axis and
    centred on the base pose: what a viewer's camera does between consecutive frames.  In the base camera's
    coordinates pose k maps x to R_y(theta_k) (x - p) + p, i.e. R' = R_y R, t' = R_y (t - p) + p."""
    R0, t0 = _rotation(qvec), np.asarray(tvec, dtype=np.float64)
    p = np.array([0.0, 0.0, float(pivot_depth)])
    out = []
    for k in range(n):
        th = np.deg2rad((k - (n - 1) / 2.0) * step_deg)
        Ry = np.array([[np.cos(th), 0.0, np.sin(th)], [0.0, 1.0, 0.0], [-np.sin(th), 0.0, np.cos(th)]])
        out.append((_qvec_from_rotation(Ry @ R0), Ry @ (t0 - p) + p))
    return out


def write_colmap_text(path: str, scene: Dict[str, np.ndarray], image_id: int = 1,
                      name: str = "synthetic.jpg", extra_poses=None) -> None:
    """Writes the two COLMAP text files a ``GaussianScene`` needs (cameras.txt, images.txt).  ``extra_poses``: more
    images of the same camera, ids ``image_id + 1 ..``, one per (qvec, tvec) -- e.g. ``orbit_poses``."""
    import os

    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "cameras.txt"), "w") as fid:
        fid.write("# Camera list with one line of data per camera:\n")
        fid.write("1 PINHOLE %d %d %r %r %r %r\n" % (
            int(scene["width"]), int(scene["height"]), float(scene["fx"]), float(scene["fy"]),
            float(scene["cx"]), float(scene["cy"])))
    with open(os.path.join(path, "images.txt"), "w") as fid:
        fid.write("# Image list with two lines of data per image:\n")
        q, t = scene["qvec"], scene["tvec"]
        fid.write("%d %r %r %r %r %r %r %r 1 %s\n" % (
            image_id, float(q[0]), float(q[1]), float(q[2]), float(q[3]),
            float(t[0]), float(t[1]), float(t[2]), name))
        fid.write("1.0 2.0 -1\n")
        for k, (q, t) in enumerate(extra_poses or []):
            fid.write("%d %r %r %r %r %r %r %r 1 pose%04d.jpg\n" % (
                image_id + 1 + k, float(q[0]), float(q[1]), float(q[2]), float(q[3]), float(t[0]), float(t[1]), float(t[2]), k))
            fid.write("1.0 2.0 -1\n")
