"""How many (Gaussian, tile) pairs of a frame could emission drop (last review, item 7)?  For every pair the reference
lists (bounding-box test, splat/gaussian_scene.py:209-218) the largest alpha = sigmoid(sigmoid(opacity)) exp(-1/2 d Q d^T)
over the tile's 256 pixel centres, from the C restatement's stage 1 (CPU only, a minute per 1M-Gaussian scene).
    python tools/attic/prunable_pairs.py [workload ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from intro_to_gaussian_splatting_amd.synthetic import make_scene, make_trained_like_scene  # noqa: E402
from oracle import c_oracle, cpu_ref  # noqa: E402

ARGS = {"c2": dict(n=100_000), "c3": dict(n=1_000_000),
        "c3_clustered": dict(n=1_000_000, cluster_fraction=0.5, cluster_area=0.05, sigma_ln=1.0), "c3_trainedlike": dict(n=1_000_000)}
W, H, TILE = 1920, 1080, 16
for wl in sys.argv[1:] or ["c3", "c3_clustered", "c3_trainedlike"]:
    a = dict(ARGS[wl])
    n = a.pop("n")
    sc = make_trained_like_scene(n, W, H, seed=0) if wl == "c3_trainedlike" else make_scene(n, W, H, seed=0, **a)
    from intro_to_gaussian_splatting_amd.image import GaussianImage  # noqa: F401  (camera constants as the scene computes them)
    import tempfile
    import torch
    from intro_to_gaussian_splatting_amd import GaussianScene, Gaussians
    from intro_to_gaussian_splatting_amd.synthetic import write_colmap_text
    with tempfile.TemporaryDirectory() as tmp:
        write_colmap_text(tmp, sc)
        g = Gaussians.from_arrays(sc["points"], sc["colors_0_255"], sc["scales"], sc["quaternions"], sc["opacity"], device="cpu")
        scene = GaussianScene(tmp, g)
    im = scene.images[1]
    c = im.gsx_camera()
    cam = cpu_ref.Camera(im.world2view.numpy(), im.full_proj_transform.numpy(), np.float32(c.tan_fovx), np.float32(c.tan_fovy),
                         np.float32(c.fx), np.float32(c.fy), c.width, c.height)
    t0 = time.time()
    pre = c_oracle.preprocess(sc["points"], g.colors.numpy(), sc["scales"], sc["quaternions"], sc["opacity"], cam)
    xy, Q = pre.points_xy.astype(np.float64), pre.inverse_covariance_2d.astype(np.float64).reshape(-1, 2, 2)
    op = 1.0 / (1.0 + np.exp(-pre.sigmoid_opacity.astype(np.float64).reshape(-1)))
    ntx, nty = len(cpu_ref.tile_origins(W, TILE)), len(cpu_ref.tile_origins(H, TILE))
    # the tiles a Gaussian is listed for: x0 in [min_x - 16, max_x] etc. (the reference's closed comparisons)
    tx0 = np.clip(np.ceil((pre.min_x.astype(np.float64) - TILE) / TILE), 0, ntx).astype(np.int64)
    tx1 = np.clip(np.floor(pre.max_x.astype(np.float64) / TILE), -1, ntx - 1).astype(np.int64)
    ty0 = np.clip(np.ceil((pre.min_y.astype(np.float64) - TILE) / TILE), 0, nty).astype(np.int64)
    ty1 = np.clip(np.floor(pre.max_y.astype(np.float64) / TILE), -1, nty - 1).astype(np.int64)
    wx, wy = np.maximum(tx1 - tx0 + 1, 0), np.maximum(ty1 - ty0 + 1, 0)
    cnt = wx * wy
    D = int(cnt.sum())
    gi = np.repeat(np.arange(cnt.size), cnt)
    off = np.arange(D) - np.repeat(np.cumsum(cnt) - cnt, cnt)
    ptx = tx0[gi] + off // np.maximum(wy[gi], 1)
    pty = ty0[gi] + off % np.maximum(wy[gi], 1)
    px = np.arange(TILE, dtype=np.float64)
    best = np.empty(D)
    for s in range(0, D, 200_000):
        e = slice(s, min(D, s + 200_000))
        k = gi[e]
        dx = (xy[k, 0][:, None] - (ptx[e] * TILE)[:, None] - px[None, :])[:, :, None]        # (pairs, 16, 1)
        dy = (xy[k, 1][:, None] - (pty[e] * TILE)[:, None] - px[None, :])[:, None, :]        # (pairs, 1, 16)
        q = Q[k]
        power = -0.5 * (q[:, 0, 0][:, None, None] * dx * dx + (q[:, 0, 1] + q[:, 1, 0])[:, None, None] * dx * dy + q[:, 1, 1][:, None, None] * dy * dy)
        best[e] = np.log2(op[k]) + power.reshape(power.shape[0], -1).max(axis=1) * np.log2(np.e)
    lens = np.bincount(ptx * nty + pty, minlength=ntx * nty)
    print("%s: D = %d pairs (tile lists: mean %.0f, max %d); largest alpha on the whole tile below 2^-26: %.1f %%, below 2^-33: %.1f %%, "
          "below 2^-40: %.1f %% of the pairs  [%.0f s]" % (wl, D, lens.mean(), lens.max(), 100 * np.mean(best < -26), 100 * np.mean(best < -33),
                                                          100 * np.mean(best < -40), time.time() - t0), flush=True)
    area = cnt[gi]
    for cap in (64, 256, 1024):
        m = area <= cap
        print("    Gaussians listed for <= %4d tiles hold %.1f %% of the pairs; pairs below 2^-40 among them: %.1f %% of ALL pairs" % (
            cap, 100 * m.mean(), 100 * np.mean(m & (best < -40))), flush=True)
    long_t = np.argsort(-lens)[:200]
    m = np.isin(ptx * nty + pty, long_t)
    print("    the 200 longest lists (%d pairs): below 2^-40 %.1f %%" % (int(m.sum()), 100 * np.mean(best[m] < -40)), flush=True)
