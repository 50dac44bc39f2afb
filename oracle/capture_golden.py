"""Golden-vector capture: runs the REFERENCE ITSELF and stores inputs + outputs as fixtures.

Runs only in the build container (it needs /root/reference, which never travels to the GPU
box).  The reference's Python is imported, never copied: this script writes *data* -- inputs
and the reference's outputs -- to tests/golden/*.npz.  Recipe: stub
the absent third-party ``plyfile`` module (only used for an IO side effect of the
``Gaussians`` constructor, splat/gaussians.py:18), write a synthetic COLMAP text model, build
``Gaussians``, overwrite its plain-tensor attributes with the fixture values, call
``GaussianScene.preprocess`` and ``GaussianScene.render_image``.

    python oracle/capture_golden.py            # all fixtures (~15 min: c1_256x256 3 min, needle 3 min, trainedlike 5 min)
    python oracle/capture_golden.py small      # only those whose name contains "small"
    python oracle/capture_golden.py tiles_     # the reference-rendered tiles of the 1M-Gaussian 1080p scenes (~10 min)
    python oracle/capture_golden.py fuzz_tiles # reference-rendered tiles of 24 random scenes (~8 min)
"""
from __future__ import annotations

import os
import sys
import tempfile
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REFERENCE = "/root/reference"
OUT_DIR = os.path.join(ROOT, "tests", "golden")

# name -> generator arguments.  tile: tile_size passed to render_image.
FIXTURES = {
    "small_64x48_n300": dict(n=300, width=64, height=48, seed=3, tile=16),
    "small_80x64_n120_tile8": dict(n=120, width=80, height=64, seed=5, tile=8),
    "cull_96x80_n400": dict(n=400, width=96, height=80, seed=7, tile=16, behind_fraction=0.25),
    "c1_256x256_n2000": dict(n=2000, width=256, height=256, seed=0, tile=16),
    # tile_size=2 as in the reference's notebooks; odd frame sizes; another camera pose
    "tile2_40x32_n80": dict(n=80, width=40, height=32, seed=11, tile=2),
    "pose_70x50_n250": dict(n=250, width=70, height=50, seed=13, tile=16,
                            qvec=(0.8, 0.3, -0.45, 0.25), tvec=(-0.7, 0.2, 2.1)),
    # many Gaussians per pixel: the T(1-alpha) < 1e-6 stop rule fires (checked by the tests)
    "dense_48x48_n1500": dict(n=1500, width=48, height=48, seed=17, tile=16),
    # points up to 2x the frustum half-width off axis with large footprints: the 1.3 tan(fov/2) clamp
    # of the EWA projection is active and off-screen Gaussians reach into the frame
    "wide_64x64_n400": dict(n=400, width=64, height=64, seed=19, tile=16, spread=2.0, sigma_scale=4.0),
    # footprints of ~0.1 px: the determinant floor 1e-3 (utils.py:387) and the discriminant floor 0.1
    # (utils.py:414) decide the conic and the radius
    "tiny_48x48_n600": dict(n=600, width=48, height=48, seed=23, tile=16, sigma_scale=0.07),
    # the notebook flow: Gaussians(points, colors) with the constructor's own scales / quaternions /
    # opacity (gaussians.py:23-33), nothing overwritten
    "defaults_64x64_n800": dict(n=800, width=64, height=64, seed=29, tile=16, defaults=True),
    # ill-conditioned footprints, ~250:1 (3 sigma = 70 .. 140 px by 0.3 .. 0.5 px): the products of d Q d^T are ~1e5
    # and cancel to a few units, so the float32 operation order of the weight decides alpha's fourth digit
    "needle_160x160_n110": dict(n=110, width=160, height=160, seed=0, tile=16, generator="needle"),
    # the statistics of a trained checkpoint: clustered, axis ratios up to 50:1, bimodal opacity
    "trainedlike_128x128_n3000": dict(n=3000, width=128, height=128, seed=0, tile=16, generator="trained"),
    # exact view-depth ties among overlapping Gaussians: what torch.argsort (unstable, gaussian_scene.py:117) does
    # with equal keys reaches the image
    "ties_64x64_n400": dict(n=400, width=64, height=64, seed=31, tile=16, generator="ties"),
    # at most three visible Gaussians: the reference's BLAS sums J @ W in another order then (probe_torch_order.py) --
    # two of nine visible (the count is only known after the cull), and a scene of three
    "fewvisible_48x48_n9": dict(n=9, width=48, height=48, seed=37, tile=16, generator="few", visible=2),
    "three_48x48_n3": dict(n=3, width=48, height=48, seed=41, tile=16),
    # ... and ONE visible Gaussian (of seven; alone): products with world2view take yet another order on one row
    "onevisible_48x48_n7": dict(n=7, width=48, height=48, seed=43, tile=16, generator="few", visible=1),
    "single_48x48_n1": dict(n=1, width=48, height=48, seed=47, tile=16),
}

# Stage 1 only (the reference's preprocess takes 0.3 .. 1.3 s at these sizes; its render_image would take days):
# the benchmark configurations C2 and C3 of BASELINE.json, 1080p, SURVEY.md section 8(d) generator, seed 0.
#   full=True   every by-original-index array of the reference's PreprocessedScene + its permutation
#   full=False  SHA-256 of each by-index array, the permutation's hash and its tie runs, the tile-list lengths
STAGE1_FIXTURES = {
    "stage1_c2_1080p_n100000": dict(n=100_000, width=1920, height=1080, seed=0, tile=16, full=True),
    "stage1_c3_1080p_n1000000": dict(n=1_000_000, width=1920, height=1080, seed=0, tile=16, full=False),
}


# Stage 2 at the METRIC's configuration, by the reference itself: whole frames are out of reach (75 us per
# (pixel, Gaussian) pair: ~23 h for C3), single tiles are not -- the tile's list built with the reference's own mask
# expressions (splat/gaussian_scene.py:209-226) from its own preprocess, composited by its own render_tile
# (splat/gaussian_scene.py:173-198): 7 .. 60 s per tile.  Which tiles: see _choose_tiles.
TILE_FIXTURES = {
    "tiles_c3_1080p_n1000000": dict(generator=None, args=dict(n=1_000_000, width=1920, height=1080, seed=0), tile=16,
                                    picks=("longest", "shortest", "first", "last", "most_tie_swaps", "most_saturated",
                                           "random", "random", "random", "random", "random", "random")),
    # BASELINE configs 2 and 4 (100k at 1080p; 5M at 4K, the "HBM-roofline stress run"): the same, so that every synthetic
    # configuration of BASELINE.json has pixels the reference itself composited
    "tiles_c2_1080p_n100000": dict(generator=None, args=dict(n=100_000, width=1920, height=1080, seed=0), tile=16,
                                   picks=("longest", "shortest", "first", "last", "most_tie_swaps", "most_saturated") + ("random",) * 10),
    "tiles_c4_4k_n5000000": dict(generator=None, args=dict(n=5_000_000, width=3840, height=2160, seed=0), tile=16,
                                 picks=("longest", "first", "last", "most_tie_swaps", "most_saturated", "random", "random")),
    "tiles_c3_clustered_1080p_n1000000": dict(generator=None, tile=16,
                                              args=dict(n=1_000_000, width=1920, height=1080, seed=0, cluster_fraction=0.5,
                                                        cluster_area=0.05, sigma_ln=1.0),
                                              picks=("longest", "ridge", "ridge", "ridge", "ridge")),
    "tiles_c3_trainedlike_1080p_n1000000": dict(generator="trained", tile=16,
                                                args=dict(n=1_000_000, width=1920, height=1080, seed=0),
                                                picks=("longest", "ridge", "ridge", "ridge", "most_saturated")),
}


def _import_reference():
    stub = types.ModuleType("plyfile")

    class PlyData:  # noqa: D401 - minimal stand-in for an absent IO dependency
        def __init__(self, *a, **k):
            pass

        def write(self, *a, **k):
            pass

        @staticmethod
        def read(*a, **k):
            raise RuntimeError("plyfile is stubbed")

    class PlyElement:
        @staticmethod
        def describe(*a, **k):
            return None

    stub.PlyData, stub.PlyElement = PlyData, PlyElement
    sys.modules["plyfile"] = stub
    sys.dont_write_bytecode = True
    sys.path.insert(0, REFERENCE)
    import torch  # noqa: F401
    from splat.gaussian_scene import GaussianScene
    from splat.gaussians import Gaussians

    return GaussianScene, Gaussians


def _generate(spec: dict) -> dict:
    from intro_to_gaussian_splatting_amd import synthetic

    spec = dict(spec)
    gen = {"needle": synthetic.make_needle_scene, "trained": synthetic.make_trained_like_scene,
           "ties": synthetic.make_tie_scene, "few": synthetic.make_few_visible_scene,
           None: synthetic.make_scene}[spec.pop("generator", None)]
    sc = gen(**spec)
    sc.pop("sh", None)              # the reference has no spherical harmonics: the base colour is what it renders
    sc.pop("sh_degree", None)
    return sc


def stage1_by_index(pre, order: np.ndarray, n: int) -> dict:
    """The reference's depth-sorted PreprocessedScene arrays put back in ORIGINAL Gaussian order (rows of culled
    Gaussians: zero), so that two implementations can be compared array by array whatever they do with equal depths."""
    out = {}
    for f in ("points_xy", "covariance_2d", "depths", "inverse_covariance_2d", "radius", "min_x", "max_x", "min_y",
              "max_y", "sigmoid_opacity", "colors"):
        a = np.ascontiguousarray(getattr(pre, f).detach().numpy() if hasattr(getattr(pre, f), "detach") else getattr(pre, f))
        full = np.zeros((n,) + a.shape[1:], dtype=a.dtype)
        full[order] = a
        out[f] = full
    return out


def sha256(a: np.ndarray) -> str:
    import hashlib

    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def tie_runs(depths_sorted: np.ndarray) -> np.ndarray:
    """Boolean mask over sorted positions: the position belongs to a run of at least two equal depths."""
    d = np.ascontiguousarray(depths_sorted, np.float32).view(np.uint32)
    same_next = np.concatenate([d[1:] == d[:-1], [False]])
    same_prev = np.concatenate([[False], d[1:] == d[:-1]])
    return same_next | same_prev


def capture_stage1(name: str, spec: dict, GaussianScene, Gaussians) -> None:
    """Stage 1 of the reference at a benchmark size: ``GaussianScene.preprocess`` itself, its permutation recovered by
    repeating its own argsort (and checked against the colours it gathered), the tile-list lengths by the reference's
    own mask expressions (splat/gaussian_scene.py:209-217)."""
    import torch

    from intro_to_gaussian_splatting_amd.synthetic import write_colmap_text

    spec = dict(spec)
    tile, full = spec.pop("tile"), spec.pop("full")
    sc = _generate(spec)
    n = sc["points"].shape[0]
    t0 = time.time()
    with tempfile.TemporaryDirectory() as tmp:
        write_colmap_text(os.path.join(tmp, "colmap"), sc)
        with torch.no_grad():
            g = Gaussians(torch.from_numpy(sc["points"]), torch.from_numpy(sc["colors_0_255"]), model_path=tmp)
            g.points = torch.from_numpy(sc["points"]).float()
            g.scales = torch.from_numpy(sc["scales"]).float()
            g.quaternions = torch.from_numpy(sc["quaternions"]).float()
            g.opacity = torch.from_numpy(sc["opacity"]).float()
            scene = GaussianScene(os.path.join(tmp, "colmap"), g)
            cam = scene.images[1]
            from splat.utils import in_view_frustum

            in_view = in_view_frustum(points=g.points, view_matrix=cam.world2view)
            t1 = time.time()
            pre = scene.preprocess(1)
            dt = time.time() - t1
            hom = torch.cat([g.points[in_view], torch.ones(int(in_view.sum()), 1)], dim=1)
            depth_unsorted = (hom @ cam.world2view)[:, 2]
            perm = torch.argsort(depth_unsorted)
            assert torch.equal(depth_unsorted[perm], pre.depths), "argsort is not reproducible"
            assert torch.equal(g.colors[in_view][perm], pre.colors), "argsort is not reproducible"
            # tile-list lengths: the reference's masks, column by column and row by row; a Gaussian is in tile
            # (tx, ty) iff both hold, so the table is the product of the two mask matrices (exact in float32: < 2^24)
            W, H = int(cam.width.item()), int(cam.height.item())
            xin = torch.stack([(pre.min_x <= x_min + tile) & (pre.max_x >= x_min) for x_min in range(0, W - tile, tile)])
            yin = torch.stack([(pre.min_y <= y_min + tile) & (pre.max_y >= y_min) for y_min in range(0, H - tile, tile)])
            counts = (xin.float() @ yin.float().T).round().to(torch.int64)
            d_ref = int((xin.sum(0).to(torch.int64) * yin.sum(0).to(torch.int64)).sum())
            assert d_ref == int(counts.sum())
    idx = np.nonzero(in_view.numpy())[0]
    order = idx[perm.numpy()].astype(np.int64)
    by_index = stage1_by_index(pre, order, n)
    ties = tie_runs(pre.depths.numpy())
    out = dict(
        generator=np.array(repr(sorted(spec.items()))), tile=np.int64(tile), n=np.int64(n), n_visible=np.int64(idx.size),
        qvec=sc["qvec"], tvec=sc["tvec"], fx=sc["fx"], fy=sc["fy"], width=sc["width"], height=sc["height"],
        inputs_sha256=np.array(sha256(np.concatenate([sc[k].reshape(-1) for k in ("points", "colors_0_255", "scales",
                                                                                   "quaternions", "opacity")]))),
        world2view=cam.world2view.numpy(), full_proj_transform=cam.full_proj_transform.numpy(),
        tan_fovX=cam.tan_fovX.numpy(), tan_fovY=cam.tan_fovY.numpy(), f_x=cam.f_x.numpy(), f_y=cam.f_y.numpy(),
        tile_instances=np.int64(d_ref), tile_counts=counts.numpy().astype(np.uint32),
        order_sha256=np.array(sha256(order.astype(np.int32))),
        # what the reference's argsort did with equal depths: the Gaussians of every run of equal depths, in ITS order
        tie_positions=np.nonzero(ties)[0].astype(np.int32), tie_order=order[ties].astype(np.int32),
        reference_preprocess_seconds=np.float64(dt),
    )
    if full:
        out.update(order=order.astype(np.int32), depths=by_index["depths"], points_xy=by_index["points_xy"],
                   covariance_2d=by_index["covariance_2d"], sigmoid_opacity=by_index["sigmoid_opacity"],
                   radius=by_index["radius"].astype(np.int16),
                   bbox=np.stack([by_index[k] for k in ("min_x", "max_x", "min_y", "max_y")], axis=1).astype(np.int16))
        for k in ("radius", "min_x", "max_x", "min_y", "max_y"):
            assert np.array_equal(by_index[k].astype(np.int16).astype(np.float32), by_index[k]), k
    for k, a in by_index.items():       # (sigmoid_opacity too: torch's SIMD sigmoid is restated, chunk tails and all)
        out["sha256_" + k] = np.array(sha256(a))
    out["torch_num_threads"] = np.int64(torch.get_num_threads())      # the sigmoid's chunking depends on it
    os.makedirs(OUT_DIR, exist_ok=True)
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print("%s: N=%d in_view=%d D=%d tied=%d ref_preprocess=%.2fs (total %.1fs) -> %s (%.0f KB)" % (
        name, n, idx.size, d_ref, int(ties.sum()), dt, time.time() - t0, path, os.path.getsize(path) / 1024))


def _final_transmittance(pre, sel, x0: int, y0: int, tile: int) -> np.ndarray:
    """Per pixel of a tile: the product of (1 - alpha) over its whole list, in float64 -- only to CHOOSE tiles (one in
    which the stop rule T (1 - alpha) < 1e-6 fires for many pixels), never compared with anything."""
    m = pre.points[sel].numpy().astype(np.float64)
    q = pre.inverse_covariance_2d[sel].numpy().astype(np.float64)
    op = 1.0 / (1.0 + np.exp(-pre.sigmoid_opacity[sel].numpy().astype(np.float64).reshape(-1)))
    px, py = np.meshgrid(np.arange(x0, x0 + tile, dtype=np.float64), np.arange(y0, y0 + tile, dtype=np.float64), indexing="ij")
    T = np.ones(px.shape)
    for k in range(m.shape[0]):
        dx, dy = m[k, 0] - px, m[k, 1] - py
        w = np.exp(-0.5 * ((dx * q[k, 0, 0] + dy * q[k, 1, 0]) * dx + (dx * q[k, 0, 1] + dy * q[k, 1, 1]) * dy))
        T = np.where(T * (1 - w * op[k]) < 1e-6, 0.0, T * (1 - w * op[k]))
    return T


def _choose_tiles(picks, pre, counts, xin, yin, swapped, tile: int, seed: int = 0):
    """Tile coordinates (tx, ty) for the named picks, from the reference's own stage 1:
      longest / shortest   the longest / shortest non-empty list of the frame
      first / last         tile (0, 0) and the last tile the reference renders (the one after it never is)
      most_tie_swaps       the list holding most Gaussians that torch.argsort put elsewhere than an index-order tie-break does
      most_saturated       among the 48 longest lists of <= 4000 entries: the one whose pixels stop earliest (float64 estimate)
      ridge                tiles along the long axis of the most ill-conditioned footprint (axis ratio from the conic:
                           rho = (1 + c) / (1 - c), c = |Q01 + Q10| / 2 sqrt(Q00 Q11)) whose weight the reference's own
                           float32 rounding moves most (8.8e-8 * opacity * rho), lists of <= 2500 entries
      random               seeded picks among lists of 0.9 .. 1.1 of the mean length"""
    rs = np.random.RandomState(seed + 2027)
    ntx, nty = counts.shape
    taken, out = set(), []

    def take(t, why):
        t = (int(t[0]), int(t[1]))
        if t in taken or counts[t] == 0:
            return False
        taken.add(t)
        out.append((t, why))
        return True

    ridge_queue = []
    if "ridge" in picks:
        q = pre.inverse_covariance_2d.numpy().astype(np.float64)
        c = np.abs(q[:, 0, 1] + q[:, 1, 0]) / (2.0 * np.sqrt(np.maximum(q[:, 0, 0] * q[:, 1, 1], 1e-300)))
        rho = (1.0 + c) / np.maximum(1.0 - c, 1e-12)
        op = 1.0 / (1.0 + np.exp(-pre.sigmoid_opacity.numpy().astype(np.float64).reshape(-1)))
        score = 8.8e-8 * op * rho
        xy = pre.points.numpy().astype(np.float64)
        inside = (xy[:, 0] > 64) & (xy[:, 0] < (ntx - 4) * tile) & (xy[:, 1] > 64) & (xy[:, 1] < (nty - 4) * tile)
        radius = pre.radius.numpy()
        for k in np.argsort(-np.where(inside & (radius >= 40) & (radius <= 400), score, 0.0))[:200]:
            # long axis of the footprint = eigenvector of the conic with the SMALL eigenvalue
            sym = np.array([[q[k, 0, 0], 0.5 * (q[k, 0, 1] + q[k, 1, 0])], [0.5 * (q[k, 0, 1] + q[k, 1, 0]), q[k, 1, 1]]])
            vals, vecs = np.linalg.eigh(sym)
            axis = vecs[:, 0]
            for step in (0.0, 20.0, -20.0, 40.0, -40.0):
                p = xy[k] + step * axis
                t = (int(p[0] // tile), int(p[1] // tile))
                if 0 <= t[0] < ntx and 0 <= t[1] < nty and counts[t] <= 2500 and bool(xin[t[0], k] & yin[t[1], k]):
                    ridge_queue.append((t, "ridge of Gaussian at sorted position %d (rho %.0f, score %.2e), %+.0f px along its long axis" % (
                        k, rho[k], score[k], step)))
    for pick in picks:
        if pick == "longest":
            take(np.unravel_index(np.argmax(counts), counts.shape), "longest list of the frame")
        elif pick == "shortest":
            take(np.unravel_index(np.argmin(np.where(counts > 0, counts, 1 << 40)), counts.shape), "shortest non-empty list")
        elif pick == "first":
            take((0, 0), "first tile")
        elif pick == "last":
            take((ntx - 1, nty - 1), "last tile the reference renders")
        elif pick == "most_tie_swaps":
            per_tile = xin[:, swapped].float() @ yin[:, swapped].float().T
            take(np.unravel_index(int(per_tile.argmax()), counts.shape), "most Gaussians whose tie order differs from index order")
        elif pick == "most_saturated":
            cand = [np.unravel_index(i, counts.shape) for i in np.argsort(-np.where(counts <= 4000, counts, 0), axis=None)[:48]]
            cand = [t for t in cand if (int(t[0]), int(t[1])) not in taken]
            stopped = [int((_final_transmittance(pre, xin[t[0]] & yin[t[1]], t[0] * tile, t[1] * tile, tile) == 0.0).sum()) for t in cand]
            best = int(np.argmax(stopped))
            take(cand[best], "%d of 256 pixels reach the stop rule (float64 estimate)" % stopped[best])
        elif pick == "ridge":
            while ridge_queue and not take(*ridge_queue.pop(0)):
                pass
        elif pick == "random":
            mean = counts[counts > 0].mean()
            ok = np.argwhere((counts > 0.9 * mean) & (counts < 1.1 * mean))
            while not take(ok[rs.randint(len(ok))], "seeded random pick among lists of about the mean length"):
                pass
    return out


def capture_tiles(name: str, spec: dict, GaussianScene, Gaussians) -> None:
    """Reference-rendered tiles of a 1M-Gaussian 1080p scene: ``GaussianScene.preprocess`` (8 torch threads, like every
    other fixture), per chosen tile the list by the reference's own mask expressions (splat/gaussian_scene.py:209-226)
    and the block by the reference's own ``render_tile`` (:173-198).  Stored: the blocks, tile coordinates, and every
    list as ORIGINAL Gaussian indices in the reference's order (so that a restatement can be given the reference's tie
    order), plus how many list positions an index-order tie-break would fill differently."""
    import torch

    from intro_to_gaussian_splatting_amd.synthetic import write_colmap_text

    tile = spec["tile"]
    sc = _generate(dict(spec["args"], generator=spec["generator"]))
    n = sc["points"].shape[0]
    t_all = time.time()
    with tempfile.TemporaryDirectory() as tmp:
        write_colmap_text(os.path.join(tmp, "colmap"), sc)
        with torch.no_grad():
            g = Gaussians(torch.from_numpy(sc["points"]), torch.from_numpy(sc["colors_0_255"]), model_path=tmp)
            g.points = torch.from_numpy(sc["points"]).float()
            g.scales = torch.from_numpy(sc["scales"]).float()
            g.quaternions = torch.from_numpy(sc["quaternions"]).float()
            g.opacity = torch.from_numpy(sc["opacity"]).float()
            scene = GaussianScene(os.path.join(tmp, "colmap"), g)
            cam = scene.images[1]
            from splat.utils import in_view_frustum

            in_view = in_view_frustum(points=g.points, view_matrix=cam.world2view)
            pre = scene.preprocess(1)
            hom = torch.cat([g.points[in_view], torch.ones(int(in_view.sum()), 1)], dim=1)
            depth_unsorted = (hom @ cam.world2view)[:, 2]
            perm = torch.argsort(depth_unsorted)
            assert torch.equal(depth_unsorted[perm], pre.depths), "argsort is not reproducible"
            assert torch.equal(g.colors[in_view][perm], pre.colors), "argsort is not reproducible"
            order = np.nonzero(in_view.numpy())[0][perm.numpy()].astype(np.int64)
            W, H = int(cam.width.item()), int(cam.height.item())
            xin = torch.stack([(pre.min_x <= x_min + tile) & (pre.max_x >= x_min) for x_min in range(0, W - tile, tile)])
            yin = torch.stack([(pre.min_y <= y_min + tile) & (pre.max_y >= y_min) for y_min in range(0, H - tile, tile)])
            counts = (xin.float() @ yin.float().T).round().to(torch.int64).numpy()
            # sorted positions that an index-order tie-break fills with another Gaussian than torch.argsort did
            ties = tie_runs(pre.depths.numpy())
            d = pre.depths.numpy().view(np.uint32)
            run = np.concatenate([[0], np.cumsum(d[1:] != d[:-1])])
            by_index = order[np.lexsort((order, run))]
            swapped = torch.from_numpy(by_index != order)
            assert not (by_index != order)[~ties].any()
            chosen = _choose_tiles(spec["picks"], pre, counts, xin, yin, swapped, tile, seed=0)
            blocks, lists, secs, swaps = [], [], [], []
            for (tx, ty), why in chosen:
                x_min, y_min = tx * tile, ty * tile
                x_in_tile = (pre.min_x <= x_min + tile) & (pre.max_x >= x_min)
                y_in_tile = (pre.min_y <= y_min + tile) & (pre.max_y >= y_min)
                points_in_tile = x_in_tile & y_in_tile
                t0 = time.time()
                block = scene.render_tile(x_min=x_min, y_min=y_min, points_in_tile_mean=pre.points[points_in_tile],
                                          colors=pre.colors[points_in_tile],
                                          opacities=pre.sigmoid_opacity[points_in_tile],
                                          inverse_covariance=pre.inverse_covariance_2d[points_in_tile], tile_size=tile)
                secs.append(time.time() - t0)
                blocks.append(block.numpy().copy())
                lists.append(order[points_in_tile.numpy()].astype(np.int32))
                swaps.append(int((swapped & points_in_tile).sum()))
                assert lists[-1].size == counts[tx, ty]
                print("  %s tile (%d, %d): list %d, %d tie-swapped, reference render_tile %.1f s -- %s" % (
                    name, tx, ty, lists[-1].size, swaps[-1], secs[-1], why), flush=True)
    out = dict(
        generator_name=np.array(spec["generator"] or "scene"), generator=np.array(repr(sorted(spec["args"].items()))),
        tile=np.int64(tile), n=np.int64(n), n_visible=np.int64(order.size),
        qvec=sc["qvec"], tvec=sc["tvec"], fx=sc["fx"], fy=sc["fy"], width=sc["width"], height=sc["height"],
        inputs_sha256=np.array(sha256(np.concatenate([sc[k].reshape(-1) for k in ("points", "colors_0_255", "scales",
                                                                                   "quaternions", "opacity")]))),
        world2view=cam.world2view.numpy(), full_proj_transform=cam.full_proj_transform.numpy(),
        tan_fovX=cam.tan_fovX.numpy(), tan_fovY=cam.tan_fovY.numpy(), f_x=cam.f_x.numpy(), f_y=cam.f_y.numpy(),
        tile_instances=np.int64(counts.sum()),
        tiles=np.array([t for t, _ in chosen], dtype=np.int32), why=np.array([w for _, w in chosen]),
        blocks=np.stack(blocks).astype(np.float32), list_len=np.array([a.size for a in lists], dtype=np.int64),
        list_indices=np.concatenate(lists), tie_swapped=np.array(swaps, dtype=np.int64),
        reference_render_tile_seconds=np.array(secs), torch_num_threads=np.int64(torch.get_num_threads()),
    )
    os.makedirs(OUT_DIR, exist_ok=True)
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print("%s: %d tiles, D=%d, %.0f s of reference render_tile (total %.0f s) -> %s (%.0f KB)" % (
        name, len(chosen), int(counts.sum()), sum(secs), time.time() - t_all, path, os.path.getsize(path) / 1024))


# Random scenes, pixels by the reference: oracle/fuzz_vs_reference.py's "tiles" cases (scenes of 3e3 .. 6e4 Gaussians from seven
# generators, random pose and frame; per scene the longest list the reference composites in ~10 s and a random one, through its
# own render_tile) for these seeds.  The scene is regenerated from the seed where the fixture is used
# (fuzz_vs_reference.random_case(seed, "tiles")): the fixture holds the reference's blocks and counts only.
FUZZ_TILE_FIXTURES = {"fuzz_tiles_seeds_8000": list(range(8000, 8024))}


def capture_fuzz_tiles(name: str, seeds, GaussianScene, Gaussians) -> None:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import fuzz_vs_reference

    rows, blocks = [], []
    for seed in seeds:
        r = fuzz_vs_reference.run_case(int(seed), "tiles", GaussianScene, Gaussians, numpy_too=False)
        assert not r["diffs"] and r["order_diffs_outside_ties"] == 0 and r["image_max_abs"] <= 2e-6, r
        for tx, ty, length, blk in r["blocks"]:
            rows.append((int(seed), r["n"], r["n_visible"], r["frame"][0], r["frame"][1], r["tile_instances"], tx, ty, length, r["tied"]))
            blocks.append(blk.astype(np.float32))
        print("  %s seed %d %-9s n=%6d %4dx%-4d D=%8d: %d tiles, lists to %d, C port (reference's order) %.1e" % (
            name, seed, r["kind"], r["n"], r["frame"][0], r["frame"][1], r["tile_instances"], len(r["blocks"]), r["longest_list"],
            r["image_max_abs"]), flush=True)
    out = dict(columns=np.array("seed n n_visible width height tile_instances tx ty list_len tied".split()),
               rows=np.array(rows, dtype=np.int64), blocks=np.stack(blocks), tile=np.int64(16))
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%d tiles of %d scenes, %.0f KB)" % (path, len(blocks), len(seeds), os.path.getsize(path) / 1024), flush=True)


def capture(name: str, spec: dict, GaussianScene, Gaussians) -> None:
    import torch

    from intro_to_gaussian_splatting_amd.synthetic import write_colmap_text

    spec = dict(spec)
    tile = spec.pop("tile")
    defaults = spec.pop("defaults", False)
    sc = _generate(spec)
    with tempfile.TemporaryDirectory() as tmp:
        write_colmap_text(os.path.join(tmp, "colmap"), sc)
        with torch.no_grad():
            g = Gaussians(torch.from_numpy(sc["points"]), torch.from_numpy(sc["colors_0_255"]), model_path=tmp)
            g.points = torch.from_numpy(sc["points"]).float()
            if defaults:        # keep what the reference's constructor set; store it as the fixture's inputs
                sc["scales"] = g.scales.detach().numpy().astype(np.float32)
                sc["quaternions"] = g.quaternions.detach().numpy().astype(np.float32)
                sc["opacity"] = g.opacity.detach().numpy().astype(np.float32)
            else:
                g.scales = torch.from_numpy(sc["scales"]).float()
                g.quaternions = torch.from_numpy(sc["quaternions"]).float()
                g.opacity = torch.from_numpy(sc["opacity"]).float()
            scene = GaussianScene(os.path.join(tmp, "colmap"), g)
            cam = scene.images[1]
            from splat.utils import in_view_frustum

            in_view = in_view_frustum(points=g.points, view_matrix=cam.world2view)
            cov3d = g.get_3d_covariance_matrix()
            pre = scene.preprocess(1)
            # the reference does not return its permutation; recover it from depths.
            hom = torch.cat([g.points[in_view], torch.ones(int(in_view.sum()), 1)], dim=1)
            depth_unsorted = (hom @ cam.world2view)[:, 2]
            perm = torch.argsort(depth_unsorted)
            assert torch.equal(depth_unsorted[perm], pre.depths), "argsort is not reproducible"
            # the debug projection helper (gaussian_scene.py:44-51 -> image.py:72-89) and the Sigma2D wrapper
            # (gaussian_scene.py:53-68) on the in-view points, in INPUT order
            pts_img, pts_col = scene.render_points_image(1)
            cov2d_wrapper = scene.get_2d_covariance(1, g.points[in_view], cov3d[in_view])
            t0 = time.time()
            image = scene.render_image(1, tile_size=tile)
            dt = time.time() - t0
    idx = np.nonzero(in_view.numpy())[0]
    out = dict(sc)
    out.update(
        tile=np.int64(tile),
        colors=g.colors.detach().numpy(),
        world2view=cam.world2view.numpy(), projection_matrix=cam.projection_matrix.numpy(),
        full_proj_transform=cam.full_proj_transform.numpy(),
        tan_fovX=cam.tan_fovX.numpy(), tan_fovY=cam.tan_fovY.numpy(),
        f_x=cam.f_x.numpy(), f_y=cam.f_y.numpy(),
        intrinsic_matrix=cam.intrinsic_matrix.numpy(), extrinsic_matrix=cam.extrinsic_matrix.numpy(),
        projection=cam.projection.numpy(), camera_center=cam.camera_center.numpy(),
        points_image_xyz=pts_img.detach().numpy(), points_image_colors=pts_col.detach().numpy(),
        get_2d_covariance=cov2d_wrapper.detach().numpy(),
        in_view=in_view.numpy(), covariance_3d=cov3d.numpy(),
        order=idx[perm.numpy()].astype(np.int64),
        pre_points=pre.points.numpy(), pre_colors=pre.colors.detach().numpy(),
        pre_covariance_2d=pre.covariance_2d.numpy(), pre_depths=pre.depths.numpy(),
        pre_inverse_covariance_2d=pre.inverse_covariance_2d.numpy(), pre_radius=pre.radius.numpy(),
        pre_points_xy=pre.points_xy.numpy(), pre_min_x=pre.min_x.numpy(), pre_min_y=pre.min_y.numpy(),
        pre_max_x=pre.max_x.numpy(), pre_max_y=pre.max_y.numpy(),
        pre_sigmoid_opacity=pre.sigmoid_opacity.numpy(),
        image=image.numpy(), reference_render_seconds=np.float64(dt),
    )
    os.makedirs(OUT_DIR, exist_ok=True)
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print("%s: N=%d in_view=%d image=%s ref_render=%.1fs -> %s (%.0f KB)" % (
        name, sc["points"].shape[0], idx.size, tuple(image.shape), dt, path, os.path.getsize(path) / 1024))


def capture_colmap_model() -> None:
    """A small COLMAP sparse model in both flavours (written here with ``struct`` / ``%r``, following the
    published COLMAP file layout) parsed by the REFERENCE's readers (splat/read_colmap.py:87-239); the
    files and what the reference read from them are the fixture for our own reader (colmap.py)."""
    import struct

    from splat import read_colmap as rc

    rs = np.random.RandomState(41)
    cams = [  # id, model id, model name, width, height, params
        (1, 1, "PINHOLE", 1959, 1090, [1159.5880733038064, 1164.6601287484507, 979.5, 545.0]),
        (2, 0, "SIMPLE_PINHOLE", 640, 480, [500.25, 320.0, 240.0]),
        (7, 4, "OPENCV", 1280, 720, [900.125, 901.5, 640.0, 360.0, -0.1, 0.01, 1e-4, -2e-4]),
        (9, 2, "SIMPLE_RADIAL", 800, 600, [610.0, 400.0, 300.0, 0.03125]),
    ]
    imgs = []
    for k, (img_id, cam_id, name, npts) in enumerate([(1, 1, "_DSC8973.JPG", 5), (2, 2, "frame 002.png", 0),
                                                       (100, 7, "sub/dir/a.jpg", 3), (31, 9, "z.JPG", 1)]):
        q = rs.normal(size=4)
        q /= np.linalg.norm(q)
        t = rs.normal(size=3) * 3.0
        xy = rs.uniform(0, 1000, (npts, 2))
        ids = rs.randint(-1, 5000, npts).astype(np.int64)
        imgs.append((img_id, q, t, cam_id, name, xy, ids))
    out_dir = os.path.join(OUT_DIR, "colmap_model")
    os.makedirs(os.path.join(out_dir, "bin"), exist_ok=True)
    os.makedirs(os.path.join(out_dir, "txt"), exist_ok=True)
    with open(os.path.join(out_dir, "bin", "cameras.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(cams)))
        for cid, mid, _, w, h, prm in cams:
            f.write(struct.pack("<iiQQ", cid, mid, w, h))
            f.write(struct.pack("<" + "d" * len(prm), *prm))
    with open(os.path.join(out_dir, "bin", "images.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(imgs)))
        for img_id, q, t, cam_id, name, xy, ids in imgs:
            f.write(struct.pack("<idddddddi", img_id, *q, *t, cam_id))
            f.write(name.replace(" ", "_").encode("utf-8") + b"\x00")
            f.write(struct.pack("<Q", len(ids)))
            for (x, y), pid in zip(xy, ids):
                f.write(struct.pack("<ddq", x, y, int(pid)))
    with open(os.path.join(out_dir, "txt", "cameras.txt"), "w") as f:
        f.write("# Camera list with one line of data per camera:\n#   CAMERA_ID, MODEL, WIDTH, HEIGHT, PARAMS[]\n")
        for cid, _, mname, w, h, prm in cams:
            f.write("%d %s %d %d %s\n" % (cid, mname, w, h, " ".join(repr(float(v)) for v in prm)))
    with open(os.path.join(out_dir, "txt", "images.txt"), "w") as f:
        f.write("# Image list with two lines of data per image:\n\n")
        for img_id, q, t, cam_id, name, xy, ids in imgs:
            f.write("%d %s %s %d %s\n" % (img_id, " ".join(repr(float(v)) for v in q),
                                          " ".join(repr(float(v)) for v in t), cam_id, name.replace(" ", "_")))
            f.write(" ".join("%r %r %d" % (float(x), float(y), int(pid)) for (x, y), pid in zip(xy, ids)) + "\n")
    parsed = {}
    for flavour, rcam, rimg, ext in (("bin", rc.read_cameras_binary, rc.read_images_binary, "bin"),
                                     ("txt", rc.read_cameras_text, rc.read_images_text, "txt")):
        cameras = rcam(os.path.join(out_dir, flavour, "cameras." + ext))
        images = rimg(os.path.join(out_dir, flavour, "images." + ext))
        parsed[flavour + "_camera_ids"] = np.array(sorted(cameras), dtype=np.int64)
        for cid, c in cameras.items():
            parsed["%s_cam%d_model" % (flavour, cid)] = np.array(c.model)
            parsed["%s_cam%d_size" % (flavour, cid)] = np.array([c.width, c.height], dtype=np.int64)
            parsed["%s_cam%d_params" % (flavour, cid)] = np.asarray(c.params, dtype=np.float64)
        parsed[flavour + "_image_ids"] = np.array(sorted(images), dtype=np.int64)
        for iid, im in images.items():
            parsed["%s_img%d_qvec" % (flavour, iid)] = np.asarray(im.qvec, dtype=np.float64)
            parsed["%s_img%d_tvec" % (flavour, iid)] = np.asarray(im.tvec, dtype=np.float64)
            parsed["%s_img%d_camera_id" % (flavour, iid)] = np.int64(im.camera_id)
            parsed["%s_img%d_name" % (flavour, iid)] = np.array(im.name)
            parsed["%s_img%d_xys" % (flavour, iid)] = np.asarray(im.xys, dtype=np.float64).reshape(-1, 2)
            parsed["%s_img%d_point3D_ids" % (flavour, iid)] = np.asarray(im.point3D_ids, dtype=np.int64)
    path = os.path.join(OUT_DIR, "colmap_model.npz")
    np.savez_compressed(path, **parsed)
    print("colmap_model: %d cameras, %d images, both flavours parsed by the reference -> %s" % (len(cams), len(imgs), path))


def main() -> None:
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    GaussianScene, Gaussians = _import_reference()
    if only in "colmap_model":
        capture_colmap_model()
    for name, spec in FIXTURES.items():
        if only in name:
            capture(name, spec, GaussianScene, Gaussians)
    for name, spec in STAGE1_FIXTURES.items():
        if only in name:
            capture_stage1(name, spec, GaussianScene, Gaussians)
    for name, spec in TILE_FIXTURES.items():
        if only in name:
            capture_tiles(name, spec, GaussianScene, Gaussians)
    for name, seeds in FUZZ_TILE_FIXTURES.items():
        if only in name:
            capture_fuzz_tiles(name, seeds, GaussianScene, Gaussians)


if __name__ == "__main__":
    main()
