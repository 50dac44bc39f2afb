"""Parity of the HIP path (through the C ABI of libgsx.so) against the oracle and the golden
vectors produced by the reference.  Needs an MI355X: ``pytest -m gpu``.

Bars (BASELINE.json north star): pixels within 1e-4 of the CPU reference; every stage-1 array
(depth, pixel position, 2D covariance, inverse, radius, bounding box) and what follows from them
(permutation outside equal depths, tile lists, instance count) bit-identical to the REFERENCE's
own output -- the kernel executes the float32 operations in the order torch executes them -- and
to the C restatement.
"""
import os

import numpy as np
import pytest
import torch

from conftest import (FUZZ_TILE_FIXTURE_NAMES, ROOT, STAGE1_FIELDS, STAGE1_NAMES, TILE_FIXTURE_NAMES,
                      assert_same_order_outside_ties, compare_stage1_with_reference, compare_tiles_with_reference,
                      fuzz_tiles_cases, golden_preprocessed, load_golden,
                      oracle_camera, rows_by_index, stage1_scene, tile_lists, tiles_scene)

pytestmark = pytest.mark.gpu

PIXEL_TOL = 1e-4


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("these tests need a GPU (run with -m gpu on an MI355X box)")


def _scene_from_arrays(tmp_path, sc, points=None):
    from intro_to_gaussian_splatting_amd import GaussianScene, Gaussians
    from intro_to_gaussian_splatting_amd.synthetic import write_colmap_text

    write_colmap_text(str(tmp_path), sc)
    g = Gaussians.from_arrays(sc["points"] if points is None else points, sc["colors_0_255"], sc["scales"],
                              sc["quaternions"], sc["opacity"], device="cuda:0")
    return GaussianScene(str(tmp_path), g)


def _scene_from_golden(tmp_path, g):
    sc = {k: g[k] for k in ("points", "colors_0_255", "scales", "quaternions", "opacity", "qvec", "tvec", "fx",
                            "fy", "cx", "cy", "width", "height")}
    return _scene_from_arrays(tmp_path, sc)


def _oracle_cam(scene, idx=1):
    from oracle import cpu_ref

    im = scene.images[idx]
    c = im.gsx_camera()
    return cpu_ref.Camera(im.world2view.cpu().numpy(), im.full_proj_transform.cpu().numpy(),
                          np.float32(c.tan_fovx), np.float32(c.tan_fovy), np.float32(c.fx), np.float32(c.fy),
                          c.width, c.height)


def _oracle_frame(scene, sc, tile=16):
    from oracle import c_oracle

    cam = _oracle_cam(scene)
    pre = c_oracle.preprocess(sc["points"], scene.gaussians.colors.cpu().numpy(), sc["scales"], sc["quaternions"],
                              sc["opacity"], cam)
    img, pairs, inst = c_oracle.render(pre, cam.width, cam.height, tile)
    return pre, img, inst


# ----------------------------------------------------------------------------- golden fixtures

def test_stage1_fields_match_reference_and_oracle(tmp_path, golden):
    _need_gpu()
    from oracle import c_oracle

    g = golden
    scene = _scene_from_golden(tmp_path, g)
    pre = scene.preprocess(1)
    order = scene.last_order.cpu().numpy().astype(np.int64)
    ref = c_oracle.preprocess(g["points"], g["colors"], g["scales"], g["quaternions"], g["opacity"], oracle_camera(g))
    assert np.array_equal(order, ref.order)
    got = {f: getattr(pre, f).cpu().numpy() for f in pre._fields}
    bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)  # noqa: E731
    # bit-identical to the C restatement ...
    for f in ("depths", "radius", "min_x", "max_x", "min_y", "max_y", "points", "covariance_2d",
              "inverse_covariance_2d", "colors"):
        assert np.array_equal(bits(got[f]), bits(getattr(ref, f))), f
    assert np.max(np.abs(got["sigmoid_opacity"] - ref.sigmoid_opacity)) <= 2.4e-7
    # ... and to the REFERENCE's own arrays, Gaussian by Gaussian (two equal depths may come in either order)
    assert_same_order_outside_ties(order, g["order"], g["pre_depths"])
    n = g["points"].shape[0]
    for f in ("depths", "radius", "min_x", "max_x", "min_y", "max_y", "points", "covariance_2d",
              "inverse_covariance_2d", "colors"):
        assert np.array_equal(bits(rows_by_index(got[f], order, n)), bits(rows_by_index(g["pre_" + f], g["order"], n))), f
    # sigmoid(opacity): torch's SIMD form, restated in the kernel -- the reference's bits except on the < 32 values at
    # the end of each of its threads' chunks, which torch gives to libm (one unit in the last place there at most)
    sa, sb = rows_by_index(got["sigmoid_opacity"], order, n), rows_by_index(g["pre_sigmoid_opacity"], g["order"], n)
    assert np.max(np.abs(sa - sb)) <= 1.2e-7 and np.count_nonzero(sa != sb) <= 31 * 8


@pytest.mark.parametrize("name", STAGE1_NAMES)
def test_stage1_at_benchmark_size_equals_the_reference_bit_for_bit(tmp_path, name):
    """BASELINE configs C2 (1e5) and C3 (1e6: the metric's configuration) at 1080p against what the reference's own
    ``GaussianScene.preprocess`` computed for them (tests/golden/stage1_*, oracle/capture_golden.py): depth, pixel
    position, 2D covariance, inverse, radius and bounding box of EVERY Gaussian bit for bit; the depth permutation equal
    to the reference's outside runs of equal depths (inside them the reference's unstable argsort follows its sort
    library, this build the original index -- the fixture records the reference's choice); and through the whole path:
    the tile lists have the reference's lengths, tile by tile, D = 422 419 / 4 219 511."""
    _need_gpu()
    g = load_golden(name)
    sc = stage1_scene(g)
    scene = _scene_from_arrays(tmp_path, sc)
    im = scene.images[1]
    assert np.array_equal(im.world2view.cpu().numpy(), g["world2view"])
    assert np.array_equal(im.full_proj_transform.cpu().numpy(), g["full_proj_transform"])
    pre = scene.preprocess(1)
    order = scene.last_order.cpu().numpy().astype(np.int64)
    fields = {f: getattr(pre, f).cpu().numpy() for f in STAGE1_FIELDS}
    report = compare_stage1_with_reference(g, fields, order)
    print("%s: %r" % (name, report))
    if "sigmoid_opacity" in g:
        sa = rows_by_index(pre.sigmoid_opacity.cpu().numpy(), order, int(g["n"]))
        assert np.max(np.abs(sa - g["sigmoid_opacity"])) <= 1.2e-7 and np.count_nonzero(sa != g["sigmoid_opacity"]) <= 31 * 8
    ntx, nty = g["tile_counts"].shape
    counts = torch.zeros(ntx * nty, dtype=torch.int32, device="cuda:0")
    st = {}
    scene.render_image_hip(1, tile_size=int(g["tile"]), tile_counts=counts, stats=st)
    assert st["n_visible"] == int(g["n_visible"]) and st["n_instances"] == int(g["tile_instances"])
    assert np.array_equal(counts.cpu().numpy().reshape(ntx, nty).astype(np.uint32), g["tile_counts"])


@pytest.mark.parametrize("name", TILE_FIXTURE_NAMES)
def test_frames_of_the_1m_scenes_hold_the_references_own_pixels(tmp_path, name):
    """BASELINE's synthetic configurations (C2 100k / 1080p, C3 1M / 1080p -- the metric's --, C4 5M / 4K) and the two 1M stress
    scenes against the reference itself, pixels included: ``tiles_*`` hold 16x16 blocks that the reference's own
    ``render_tile`` (splat/gaussian_scene.py:173-198) composited from its own ``preprocess`` of those scenes -- the frame's
    longest list (85 / 642 / 809 / 12 061 / 10 298 entries), tiles
    where the stop rule fires, tiles along the ridge of a 330:1 footprint (the kernel's reference-order records), seeded
    random picks.  The HIP frame (whole path, one call, the kernels bench.py times) carries those pixels to 1e-4 (the
    north-star tolerance; measured <= 1e-6), its tile lists have the reference's lengths, and so does a captured frame."""
    _need_gpu()
    g = load_golden(name)
    sc = tiles_scene(g)
    scene = _scene_from_arrays(tmp_path, sc)
    w, h, t = int(g["width"]), int(g["height"]), int(g["tile"])
    ntx, nty = (w - 1) // t, (h - 1) // t
    counts = torch.zeros(ntx * nty, dtype=torch.int32, device="cuda:0")
    st = {}
    img = scene.render_image_hip(1, tile_size=t, tile_counts=counts, stats=st)
    assert st["n_visible"] == int(g["n_visible"]) and st["n_instances"] == int(g["tile_instances"])
    counts = counts.cpu().numpy().reshape(ntx, nty)
    for k, (tx, ty) in enumerate(g["tiles"]):
        assert counts[tx, ty] == g["list_len"][k]
    rep = compare_tiles_with_reference(g, img.cpu().numpy())
    print("%s: %r" % (name, rep))
    assert rep["max_abs"] <= PIXEL_TOL and rep["tiles"] == len(g["tiles"])
    assert rep["max_abs"] <= 2e-6           # measured; the tolerance the north star states is the line above
    frame = scene.capture_frame(1, tile_size=t)
    rep2 = compare_tiles_with_reference(g, frame.replay().cpu().numpy())
    frame.confirm()
    assert rep2["per_tile"] == rep["per_tile"]


def test_equal_depths_follow_the_original_index_where_the_reference_follows_its_sort_library(tmp_path):
    """ties_64x64_n400 (367 of 400 Gaussians share their depth with another): the kernel's stage 1 equals the
    reference's Gaussian by Gaussian, its permutation differs from the reference's only inside runs of equal depths,
    and the frame equals the C restatement's -- which, GIVEN the reference's permutation, reproduces the reference's
    image (tests/test_oracle_golden.py).  What the tie order does to the picture is measured there."""
    _need_gpu()
    g = load_golden("ties_64x64_n400")
    scene = _scene_from_golden(tmp_path, g)
    pre = scene.preprocess(1)
    order = scene.last_order.cpu().numpy().astype(np.int64)
    differing = assert_same_order_outside_ties(order, g["order"], g["pre_depths"])
    assert differing > 50
    n = g["points"].shape[0]
    for f in ("depths", "radius", "min_x", "max_x", "min_y", "max_y", "points", "covariance_2d", "inverse_covariance_2d"):
        a, b = rows_by_index(getattr(pre, f).cpu().numpy(), order, n), rows_by_index(g["pre_" + f], g["order"], n)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f
    sc = {k: g[k] for k in ("points", "scales", "quaternions", "opacity")}
    ref_pre, ref, inst = _oracle_frame(scene, sc, int(g["tile"]))
    assert np.array_equal(ref_pre.order, order)
    st = {}
    img = scene.render_image_hip(1, tile_size=int(g["tile"]), stats=st).cpu().numpy()
    assert st["n_instances"] == inst and np.max(np.abs(img - ref)) <= 1e-5
    # with the reference's own permutation (its stage-1 arrays through the native boundary): the reference's image
    from intro_to_gaussian_splatting_amd import render_preprocessed

    t = lambda k: torch.from_numpy(np.ascontiguousarray(g["pre_" + k])).to("cuda:0")  # noqa: E731
    theirs = render_preprocessed(int(g["height"]), int(g["width"]), int(g["tile"]), t("points"), t("colors"),
                                 t("inverse_covariance_2d"), t("min_x"), t("max_x"), t("min_y"), t("max_y"), t("sigmoid_opacity"))
    assert np.max(np.abs(theirs.cpu().numpy() - g["image"])) <= 1e-5


def test_blend_given_the_references_stage1_arrays(golden):
    """Stage-wise parity: the reference's own PreprocessedScene arrays go through the native
    boundary (gsx_render_preprocessed mirrors splat/c/render.cu:90-101)."""
    _need_gpu()
    from intro_to_gaussian_splatting_amd import render_preprocessed

    g = golden
    dev = "cuda:0"
    t = lambda k: torch.from_numpy(np.ascontiguousarray(g["pre_" + k])).to(dev)  # noqa: E731
    stats = {}
    img = render_preprocessed(int(g["height"]), int(g["width"]), int(g["tile"]), t("points"), t("colors"),
                              t("inverse_covariance_2d"), t("min_x"), t("max_x"), t("min_y"), t("max_y"),
                              t("sigmoid_opacity"), stats=stats)
    assert tuple(img.shape) == g["image"].shape
    assert np.max(np.abs(img.cpu().numpy() - g["image"])) <= PIXEL_TOL
    from oracle import c_oracle

    _, _, inst = c_oracle.render(golden_preprocessed(g), int(g["width"]), int(g["height"]), int(g["tile"]))
    assert stats["n_instances"] == inst


@pytest.mark.parametrize("long_tile", [False, True])
def test_a_monomial_record_beside_reference_order_records(long_tile):
    """The stage-2 entry with a conic the completed square does not exist for (Q11 = 0: the monomial fallback) in the same
    batches as the needles' reference-order records.  A batch's kind is then "monomial" whatever else it holds: the
    launch a frame runs still counts its tiles in n_redo, and the plain instance (GSX_FLAG_PLAIN_FOOTPRINTS) still leaves
    them undone and says so -- round 5 composited the flagged records by the completed square there and reported 0.
    ``long_tile``: the lists repeated until the tiles are split over four waves (the quarter kernel)."""
    _need_gpu()
    from intro_to_gaussian_splatting_amd import _ffi, render_preprocessed
    from oracle import c_oracle, cpu_ref

    g = load_golden("needle_160x160_n110")
    w, h = int(g["width"]), int(g["height"])
    pre = {k: np.ascontiguousarray(g["pre_" + k]).copy() for k in ("points", "colors", "inverse_covariance_2d", "min_x",
                                                                   "max_x", "min_y", "max_y", "sigmoid_opacity")}
    n0 = pre["points"].shape[0]
    for k in range(0, n0, 9):         # every ninth row: a vertical bar, weight exp(-q00 dx^2 / 2), in every tile of its columns
        pre["inverse_covariance_2d"][k] = np.array([[0.02, 0.0], [0.0, 0.0]], np.float32)
        pre["min_y"][k], pre["max_y"][k] = -1.0, float(h)
    if long_tile:    # 23 fainter copies of everything, binned (by their caller-given boxes) into tile (4, 4) alone: a list of
        # > 2 500 entries, > 4x the frame's average -- the tile is split over four waves
        more = {k: np.concatenate([v] * 23) for k, v in pre.items()}
        more["sigmoid_opacity"] = (more["sigmoid_opacity"] - 3.0).astype(np.float32)
        for k in ("min_x", "min_y"):
            more[k][:] = 65.0
        for k in ("max_x", "max_y"):
            more[k][:] = 79.0
        pre = {k: np.concatenate([pre[k], more[k]]) for k in pre}
    n = pre["points"].shape[0]
    ref_pre = cpu_ref.Preprocessed(pre["points"], pre["colors"], np.zeros((n, 2, 2), np.float32), np.zeros(n, np.float32),
                                   pre["inverse_covariance_2d"], np.zeros(n, np.float32), pre["points"], pre["min_x"],
                                   pre["min_y"], pre["max_x"], pre["max_y"], pre["sigmoid_opacity"], np.arange(n))
    ref, _, inst = c_oracle.render(ref_pre, w, h, 16)
    t = {k: torch.from_numpy(v).to("cuda:0") for k, v in pre.items()}
    args = (h, w, 16, t["points"], t["colors"], t["inverse_covariance_2d"], t["min_x"], t["max_x"], t["min_y"], t["max_y"],
            t["sigmoid_opacity"])
    st = {}
    img = render_preprocessed(*args, stats=st)
    assert st["n_instances"] == inst
    assert np.max(np.abs(img.cpu().numpy() - ref)) <= 1e-5
    assert st["n_redo"] > 0                         # tiles with reference-order records, counted although their batches are "monomial"
    plain = {}
    render_preprocessed(*args, stats=plain, flags=_ffi.GSX_FLAG_PLAIN_FOOTPRINTS)
    assert plain["n_redo"] == st["n_redo"]          # the same tiles / quarters: left undone, and reported


def test_full_path_matches_reference_image(tmp_path, golden):
    _need_gpu()
    g = golden
    scene = _scene_from_golden(tmp_path, g)
    stats = {}
    img = scene.render_image_hip(1, tile_size=int(g["tile"]), stats=stats)
    assert tuple(img.shape) == (int(g["width"]), int(g["height"]), 3)      # [x, y] like the reference
    err = np.max(np.abs(img.cpu().numpy() - g["image"]))
    assert err <= PIXEL_TOL, err
    assert stats["n_visible"] == int(g["in_view"].sum())
    again = scene.render_image(1, tile_size=int(g["tile"]))
    assert again.device.type == "cpu" and torch.equal(img.cpu(), again)     # deterministic; host tensor like the reference


@pytest.mark.parametrize("name", ["fewvisible_48x48_n9", "onevisible_48x48_n7", "three_48x48_n3", "c1_256x256_n2000"])
def test_captured_frame_keeps_the_row_class_of_its_view(tmp_path, name):
    """A captured call cannot be issued again when n_visible says it assumed the wrong row class (GSX_FLAG_SMALL_BATCH /
    _ONE_VISIBLE): capture_frame bakes in the class the view's uncaptured frame ended up with -- the replay equals the
    uncaptured frame bit for bit, and the reference's image to the usual tolerance."""
    _need_gpu()
    g = load_golden(name)
    scene = _scene_from_golden(tmp_path, g)
    direct = scene.render_image_hip(1, tile_size=int(g["tile"])).clone()
    frame = scene.capture_frame(1, tile_size=int(g["tile"]))
    for _ in range(2):
        img = frame.replay()
        torch.cuda.synchronize()
        assert torch.equal(img, direct)
    assert np.max(np.abs(img.cpu().numpy() - g["image"])) <= PIXEL_TOL
    frame.confirm()


def test_tile_size_two_like_the_notebook(tmp_path):
    """cpu_render.ipynb:161 renders with tile_size=2; membership lists depend on the tile size."""
    _need_gpu()
    g = load_golden("small_80x64_n120_tile8")
    scene = _scene_from_golden(tmp_path, g)
    sc = {k: g[k] for k in ("points", "scales", "quaternions", "opacity")}
    for tile in (1, 2, 5, 32, 63, 64):                  # 63 / 64: one tile (or none) covers the frame
        _, ref, inst = _oracle_frame(scene, sc, tile)
        stats = {}
        img = scene.render_image_hip(1, tile_size=tile, stats=stats)
        assert stats["n_instances"] == inst
        assert np.max(np.abs(img.cpu().numpy() - ref)) <= PIXEL_TOL


# ----------------------------------------------------------------------------- edge cases

def test_empty_all_culled_and_tiny_frames(tmp_path):
    _need_gpu()
    from intro_to_gaussian_splatting_amd.synthetic import make_scene

    sc = make_scene(64, 64, 48, seed=4)
    empty = {k: (v[:0] if isinstance(v, np.ndarray) and v.ndim == 2 else v) for k, v in sc.items()}
    scene = _scene_from_arrays(tmp_path / "e", empty)
    img = scene.render_image(1)
    assert tuple(img.shape) == (64, 48, 3) and not img.any()
    assert scene.preprocess(1).points.shape[0] == 0
    # everything behind the camera
    sc_b = make_scene(64, 64, 48, seed=4, behind_fraction=1.0)
    scene = _scene_from_arrays(tmp_path / "b", sc_b)
    stats = {}
    img = scene.render_image_hip(1, stats=stats)
    assert not img.any() and stats["n_visible"] == 0 and stats["n_instances"] == 0
    # a frame no larger than one tile renders nothing (range(0, W - tile, tile) is empty)
    sc_t = make_scene(64, 16, 16, seed=4)
    scene = _scene_from_arrays(tmp_path / "t", sc_t)
    assert not scene.render_image(1).any()
    # an empty scene right after a real frame: its counts are zeros, not what the shared workspace held
    _scene_from_arrays(tmp_path / "w", sc).render_image_hip(1)
    stats = {}
    _scene_from_arrays(tmp_path / "e2", empty).render_image_hip(1, stats=stats)
    assert stats["n_visible"] == 0 and stats["n_instances"] == 0


def test_last_tile_row_and_column_stay_zero_and_single_splat_value(tmp_path):
    _need_gpu()
    from intro_to_gaussian_splatting_amd.synthetic import make_scene

    sc = make_scene(1, 64, 64, seed=9)
    # one big isotropic opaque splat at the image centre
    sc["scales"][:] = 0.5
    sc["quaternions"][:] = [1, 0, 0, 0]
    sc["opacity"][:] = 4.0
    sc["colors_0_255"][:] = [256.0, 128.0, 64.0]
    scene = _scene_from_arrays(tmp_path, sc)
    pre = scene.preprocess(1)
    img = scene.render_image(1).cpu().numpy()
    assert np.all(img[48:] == 0) and np.all(img[:, 48:] == 0)          # 64 - 16: never rendered
    x, y = [int(round(v)) for v in pre.points[0].cpu().numpy()]
    q = pre.inverse_covariance_2d[0].cpu().numpy().astype(np.float64)
    d = pre.points[0].cpu().numpy().astype(np.float64) - np.array([x, y], np.float64)
    w = np.exp(-0.5 * d @ q @ d)
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))  # noqa: E731
    alpha = w * sig(sig(4.0))                                           # double sigmoid
    if x < 48 and y < 48:
        assert np.allclose(img[x, y], alpha * np.array([1.0, 0.5, 0.25]), atol=2e-6)


def test_depth_order_matters_and_ties_follow_the_original_index(tmp_path):
    _need_gpu()
    from intro_to_gaussian_splatting_amd.synthetic import make_scene

    sc = make_scene(400, 96, 80, seed=11)
    # duplicate every point: exact depth ties, different colours -> order-sensitive result
    for k in ("points", "scales", "quaternions", "opacity", "colors_0_255"):
        sc[k] = np.concatenate([sc[k], sc[k]], axis=0)
    sc["colors_0_255"][400:] = sc["colors_0_255"][:400][:, ::-1]
    scene = _scene_from_arrays(tmp_path, sc)
    pre_o, ref, inst = _oracle_frame(scene, sc)
    scene.preprocess(1)
    assert np.array_equal(scene.last_order.cpu().numpy().astype(np.int64), pre_o.order)
    img = scene.render_image(1).cpu().numpy()
    assert np.max(np.abs(img - ref)) <= PIXEL_TOL
    # reversing the tie order changes pixels by far more than the tolerance
    swapped = dict(sc)
    for k in ("points", "scales", "quaternions", "opacity", "colors_0_255"):
        swapped[k] = np.concatenate([sc[k][400:], sc[k][:400]], axis=0)
    scene2 = _scene_from_arrays(tmp_path / "s", swapped)
    assert np.max(np.abs(scene2.render_image(1).cpu().numpy() - img)) > 100 * PIXEL_TOL


def test_huge_and_offscreen_splats(tmp_path):
    """Bounding boxes far outside the frame, a splat covering every tile, NaN inputs."""
    _need_gpu()
    from intro_to_gaussian_splatting_amd.synthetic import make_scene

    sc = make_scene(300, 128, 96, seed=13)
    sc["scales"][0] = 5.0                       # covers the whole frame: wave-cooperative emit path
    sc["scales"][1] = 1e-7                      # determinant floor / eigen floor
    sc["points"][2] = sc["points"][2] * 50.0    # far off screen
    scene = _scene_from_arrays(tmp_path, sc)
    _, ref, inst = _oracle_frame(scene, sc)
    stats = {}
    img = scene.render_image_hip(1, stats=stats).cpu().numpy()
    assert stats["n_instances"] == inst
    assert np.max(np.abs(img - ref)) <= PIXEL_TOL


# ----------------------------------------------------------------------------- layouts, windows

def test_layouts_and_tile_windows_are_bit_identical(tmp_path):
    _need_gpu()
    g = load_golden("c1_256x256_n2000")
    scene = _scene_from_golden(tmp_path, g)
    full = scene.render_image_hip(1, layout="wh3")
    hw = scene.render_image_hip(1, layout="hw3")
    assert tuple(hw.shape) == (256, 256, 3)
    assert torch.equal(hw.permute(1, 0, 2), full)
    # column strips: windows of the leading axis, strip-sized output buffers
    pieces = []
    for t0, t1 in ((0, 4), (4, 9), (9, 15)):
        strip = torch.full(((t1 - t0) * 16, 256, 3), -1.0, device="cuda:0")
        scene.render_image_hip(1, layout="wh3", tile_window=(t0, t1, 0, 15), out=strip, out_origin=(t0 * 16, 0))
        pieces.append(strip)
    assert torch.equal(torch.cat(pieces, 0), full[: 15 * 16])
    # a window in both axes leaves everything else zero
    part = scene.render_image_hip(1, layout="wh3", tile_window=(3, 6, 2, 5))
    assert torch.equal(part[48:96, 32:80], full[48:96, 32:80])
    part[48:96, 32:80] = 0
    assert not part.any()


def test_strip_sharding_single_rank_equals_plain_render(tmp_path):
    _need_gpu()
    from intro_to_gaussian_splatting_amd import strips

    g = load_golden("cull_96x80_n400")
    scene = _scene_from_golden(tmp_path, g)
    for layout in ("wh3", "hw3"):
        def fn(window, out, origin, layout=layout):
            scene.render_image_hip(1, layout=layout, tile_window=window, out=out, out_origin=origin)
        frame = strips.render_sharded(fn, 96, 80, 16, layout, torch.device("cuda:0"))
        assert torch.equal(frame, scene.render_image_hip(1, layout=layout))


# ----------------------------------------------------------------------------- full sizes

@pytest.mark.parametrize("n,width,height", [(100_000, 1920, 1080), (1_000_000, 1920, 1080)])
def test_full_size_frames_against_the_c_oracle(tmp_path, n, width, height):
    """BASELINE configs C2 / C3 (synthetic): whole frame vs the multi-threaded C restatement."""
    _need_gpu()
    from intro_to_gaussian_splatting_amd.synthetic import make_scene

    sc = make_scene(n, width, height, seed=0)
    scene = _scene_from_arrays(tmp_path, sc)
    pre_o, ref, inst = _oracle_frame(scene, sc)
    stats = {}
    img = scene.render_image_hip(1, stats=stats)
    assert stats["n_visible"] == pre_o.points.shape[0]
    assert stats["n_instances"] == inst
    img = img.cpu().numpy()
    err = np.max(np.abs(img - ref))
    assert err <= PIXEL_TOL, err
    # size-independent properties
    t = 16
    from intro_to_gaussian_splatting_amd.strips import tiles_along

    rx, ry = tiles_along(width, t) * t, tiles_along(height, t) * t       # 1904 x 1072 at 1080p
    assert np.all(img[rx:] == 0) and np.all(img[:, ry:] == 0) and rx < width and ry < height
    assert img[:rx, :ry].max() > 0.1
    assert img.min() >= 0.0 and img.max() < 1.0           # sum of T*alpha*c with c < 1 never reaches 1
    pre = scene.preprocess(1)
    d = pre.depths.cpu().numpy()
    assert np.all(d[1:] >= d[:-1]) and d[0] >= 0.2          # sortedness + cull plane
    assert np.array_equal(scene.last_order.cpu().numpy().astype(np.int64), pre_o.order)


def test_4k_5m_stress_properties(tmp_path):
    """BASELINE config C4 (5M Gaussians, 3840x2160): determinism, strips == frame, zero border;
    pixel parity on a window the oracle renders in seconds."""
    _need_gpu()
    from intro_to_gaussian_splatting_amd.synthetic import make_scene
    from oracle import c_oracle

    sc = make_scene(5_000_000, 3840, 2160, seed=0)
    scene = _scene_from_arrays(tmp_path, sc)
    stats = {}
    a = scene.render_image_hip(1, stats=stats)
    b = scene.render_image_hip(1)
    assert torch.equal(a, b)
    assert not a[3840 - 16:].any() and not a[:, 2160 - 16:].any()
    half = scene.render_image_hip(1, tile_window=(100, 140, 0, 134))
    assert torch.equal(half[1600:2240], a[1600:2240])
    cam = _oracle_cam(scene)
    pre = c_oracle.preprocess(sc["points"], scene.gaussians.colors.cpu().numpy(), sc["scales"], sc["quaternions"],
                              sc["opacity"], cam)
    assert stats["n_visible"] == pre.points.shape[0]
    win, _, inst = c_oracle.render(pre, 3840, 2160, 16, window=(100, 110, 60, 70))
    assert stats["n_instances"] == inst
    got = a[1600:1760, 960:1120].cpu().numpy()
    assert np.max(np.abs(got - win[1600:1760, 960:1120])) <= PIXEL_TOL
    # the per-tile list lengths the frame reports add up to D, and strips balanced by them (8 ranks'
    # worth, rendered one after the other on this GPU) assemble to the same frame bit for bit
    from intro_to_gaussian_splatting_amd import strips

    ntx, nty = strips.tiles_along(3840, 16), strips.tiles_along(2160, 16)
    counts = torch.zeros(ntx * nty, dtype=torch.int32, device="cuda:0")
    c = scene.render_image_hip(1, tile_counts=counts)
    assert torch.equal(c, a) and int(counts.sum().item()) == inst
    plan = strips.balanced_plan(strips.tile_row_costs(counts, ntx, nty), 8)
    assert plan[0][0] == 0 and plan[-1][1] == ntx and len({b - a_ for a_, b in plan}) > 1      # not the equal split
    frame = torch.full_like(a, 7.0)
    frame[ntx * 16:].zero_()
    per_rank = []
    for t0, t1 in plan:
        # twice: the second frame of a window is what a rank renders frame after frame -- pair capacity sized by the
        # count, depth-sort route picked from the kept count (GsxParams.kept_hint), splitters and schedule from the
        # hints the first one left
        for rep in range(2):
            st = {}
            scene.render_image_hip(1, tile_window=(t0, t1, 0, nty), out=frame[t0 * 16:t1 * 16], out_origin=(t0 * 16, 0),
                                   stats=st, timing=True)
        assert st["n_kept"] < 0.2 * 5_000_000                 # an eighth of the frame keeps about an eighth of the Gaussians
        per_rank.append((st["n_instances"], st["stage_ms"]))
    assert torch.equal(frame, a)
    assert sum(p[0] for p in per_rank) >= inst               # a Gaussian on a strip border is binned by both ranks
    worst = max(p[0] for p in per_rank)
    assert worst <= 1.15 * inst / 8 + 0.02 * inst            # balanced: no rank far above its share
    # the replicated share of a rank: projection of all N + ONE pass over N keys; everything after it is M ~ N/8
    full_ms = stats_timing = None
    st = {}
    scene.render_image_hip(1, stats=st, timing=True)
    full_ms = st["stage_ms"]
    strip_ms = per_rank[3][1]
    print("C4 full frame stage ms: %s" % {k: round(v, 3) for k, v in full_ms.items()})
    print("C4 1/8 strip stage ms:  %s" % {k: round(v, 3) for k, v in strip_ms.items()})
    assert strip_ms["depth_sort"] + strip_ms["scan"] <= 0.5 * (full_ms["depth_sort"] + full_ms["scan"])
    # round-3 bar (VERDICT r2, next #1b): a rank's strip in <= 0.40 ms where the frame takes ~2 ms -- measured as the
    # median of event-bracketed frames by `bench.py --workload c4 --strip-of 8` (0.386 ms); the per-stage times printed
    # here are taken with a host synchronisation per stage mark and add up to more
    assert strip_ms["total"] <= 0.30 * full_ms["total"]


@pytest.mark.parametrize("name", FUZZ_TILE_FIXTURE_NAMES)
def test_reference_rendered_tiles_of_random_scenes_on_the_gpu(name, tmp_path):
    """tests/golden/fuzz_tiles_* (24 random scenes from seven generators, 47 tiles composited by the REFERENCE's own render_tile,
    lists to 1 291 entries; oracle/capture_golden.py: capture_fuzz_tiles): the HIP frame of every scene -- whole path, one call --
    carries the reference's counts and its pixels (bar 1e-4; expected 1e-6 and below)."""
    _need_gpu()
    g = load_golden(name)
    worst, tiles = 0.0, 0
    for seed, sc, row, blocks in fuzz_tiles_cases(g):
        scene = _scene_from_arrays(tmp_path / str(seed), sc)
        st = {}
        img = scene.render_image_hip(1, stats=st).cpu().numpy()
        assert st["n_visible"] == row["n_visible"] and st["n_instances"] == row["tile_instances"], (seed, st, row)
        for tx, ty, length, blk in blocks:
            d = float(np.abs(img[tx * 16:(tx + 1) * 16, ty * 16:(ty + 1) * 16] - blk).max())
            worst, tiles = max(worst, d), tiles + 1
            assert d <= PIXEL_TOL, (seed, tx, ty, length, d)
    print("%s: %d reference-rendered tiles of %d scenes, HIP max |dpixel| %.2e" % (name, tiles, len(set(g["rows"][:, 0])), worst))


def test_kernels_stay_inside_the_buffers_they_are_given(tmp_path):
    """No GPU address sanitizer runs on this pool, so the bounds are checked the old way: workspace, hints buffer and output frame
    are handed over as the MIDDLE of larger allocations whose margins hold a pattern, the workspace at exactly the size
    gsx_workspace_bytes() names for the pair capacity -- the capacity the frame needs to the pair, one pair more, half of it
    (GSX_ERR_WORKSPACE_TOO_SMALL: pairs are dropped, nothing may be written behind the end), a single pair -- over every depth-sort
    route (one workgroup / 256 buckets / four LSD passes), long tiles on helper waves, other tile sizes, a tile window, both
    layouts, frames with and without hints.  After every frame the margins still hold the pattern."""
    _need_gpu()
    from intro_to_gaussian_splatting_amd import _ffi
    from intro_to_gaussian_splatting_amd.synthetic import make_scene

    lib = _ffi.load()
    margin = 1 << 20

    def guarded(nbytes, fill=0):
        big = torch.full((margin + nbytes + margin,), 0xA5, dtype=torch.uint8, device="cuda:0")
        big[margin:margin + nbytes] = fill
        return big, big[margin:margin + nbytes]

    def intact(big, nbytes):
        return bool((big[:margin] == 0xA5).all().item()) and bool((big[margin + nbytes:] == 0xA5).all().item())

    cases = [  # (scene arguments, frame, tile, layout, window)
        (dict(n=9_000, seed=31), (320, 240), 16, "wh3", None),                                    # one-workgroup depth sort
        (dict(n=200_000, seed=32), (1280, 720), 16, "hw3", None),                                 # 256 buckets
        (dict(n=200_000, seed=32), (1280, 720), 16, "wh3", (7, 41, 3, 29)),                       # ... on a tile window
        (dict(n=150_000, seed=33, cluster_fraction=0.5, cluster_area=0.05, sigma_ln=1.0), (1920, 1080), 16, "wh3", None),   # long tiles
        (dict(n=2_000_000, seed=34, sigma_scale=0.5), (1920, 1080), 16, "wh3", None),             # four LSD passes (> 1.5M kept)
        (dict(n=60_000, seed=35), (640, 480), 8, "wh3", None),                                    # the any-tile-size kernels
        (dict(n=60_000, seed=35), (333, 777), 5, "hw3", None),
    ]
    checked = 0
    for k, (args, (w, h), tile, layout, window) in enumerate(cases):
        sc = make_scene(width=w, height=h, **args)
        scene = _scene_from_arrays(tmp_path / str(k), sc)
        n = sc["points"].shape[0]
        st = {}
        want = scene.render_image_hip(1, tile_size=tile, layout=layout, tile_window=window, stats=st).clone()
        d = int(st["n_instances"])
        hbytes = lib.gsx_hints_bytes(w, h, tile)
        frame_bytes = want.numel() * 4
        for cap in (d, d + 1, max(d // 2, 1), 1):
            nbytes = lib.gsx_workspace_bytes(n, w, h, tile, cap)
            assert nbytes > 0
            ws_big, ws = guarded(nbytes, fill=0x5A)
            hints_big, hints = guarded(hbytes)
            out_big, out_bytes = guarded(frame_bytes)
            out = out_bytes.view(torch.float32).view(want.shape)
            for rep in range(3):            # the first frame fills the hints buffer, the next ones use it
                private = dict(cap=cap, workspace=ws, hints=[hints, rep > 0])
                try:
                    got = scene.render_image_hip(1, tile_size=tile, layout=layout, tile_window=window, out=out, _private=private)
                    torch.cuda.synchronize()
                    assert cap >= d, ("a workspace for %d pairs took %d without an error" % (cap, d), k)
                    assert torch.equal(got, want), (k, cap, rep)
                except _ffi.GsxError as exc:
                    torch.cuda.synchronize()
                    assert cap < d and exc.code == _ffi.GSX_ERR_WORKSPACE_TOO_SMALL, (k, cap, d, str(exc))
                assert intact(ws_big, nbytes), ("workspace margins overwritten", k, cap, rep)
                assert intact(hints_big, hbytes), ("hints margins overwritten", k, cap, rep)
                assert intact(out_big, frame_bytes), ("frame margins overwritten", k, cap, rep)
                checked += 1
    print("%d frames inside guarded buffers: margins intact" % checked)


def test_twenty_million_gaussians_through_the_same_path(tmp_path):
    """Sizes beyond the BASELINE configurations (a 288 GB part holds scenes of hundreds of millions of Gaussians; tools/big_scene.py
    ran 150M / 293M pairs against the whole C frame, profiles/r6_big_scenes.txt): 20M Gaussians at 1080p -- 19.8M kept, the
    4-pass route of the depth sort far above its 1.5M threshold, 50M+ pairs -- with the counts of the C restatement, pixel
    parity on a window it renders in seconds, and eight strips that assemble to the frame bit for bit."""
    _need_gpu()
    from intro_to_gaussian_splatting_amd.synthetic import make_scene
    from oracle import c_oracle

    n, w, h = 20_000_000, 1920, 1080
    sc = make_scene(n, w, h, seed=3, sigma_scale=0.45)
    scene = _scene_from_arrays(tmp_path, sc)
    st = {}
    a = scene.render_image_hip(1, stats=st)
    pre = c_oracle.preprocess(sc["points"], scene.gaussians.colors.cpu().numpy(), sc["scales"], sc["quaternions"], sc["opacity"],
                              _oracle_cam(scene))
    win, _, inst = c_oracle.render(pre, w, h, 16, window=(40, 48, 20, 28))
    print("20M: visible %d, kept %d, pairs %d" % (st["n_visible"], st["n_kept"], st["n_instances"]))
    assert st["n_visible"] == pre.points.shape[0] and st["n_instances"] == inst and inst > n
    got = a[640:768, 320:448].cpu().numpy()
    assert np.max(np.abs(got - win[640:768, 320:448])) <= PIXEL_TOL
    ntx, nty = (w + 15) // 16, (h + 15) // 16
    parts = torch.full_like(a, 3.0)
    for k in range(8):
        c0, c1 = ntx * k // 8, ntx * (k + 1) // 8
        scene.render_image_hip(1, tile_window=(c0, c1, 0, nty), out=parts[c0 * 16:min(c1 * 16, w)], out_origin=(c0 * 16, 0))
    assert torch.equal(parts, a)


def test_long_tiles_on_four_waves_equal_the_single_wave_path(tmp_path):
    """Heavy-tailed scene (half of the Gaussians inside 5 % of the frame, footprint sigma_ln 1.0): the longest
    tile lists are ~20x the mean.  Tiles above long_tile_threshold are composited by four waves, one pixel per
    lane, eight records per trip; the frame must equal the one-wave-per-tile frame bit for bit (both layouts),
    the count of such tiles must be what the per-tile list lengths imply, and the pixels must match the oracle."""
    _need_gpu()
    from intro_to_gaussian_splatting_amd import strips
    from intro_to_gaussian_splatting_amd.synthetic import make_scene
    from oracle import c_oracle

    w, h = 640, 400
    sc = make_scene(150_000, w, h, seed=4, cluster_fraction=0.5, cluster_area=0.05, sigma_ln=1.0)
    scene = _scene_from_arrays(tmp_path, sc)
    ntx, nty = strips.tiles_along(w, 16), strips.tiles_along(h, 16)
    counts = torch.zeros(ntx * nty, dtype=torch.int32, device="cuda:0")
    st = {}
    split = scene.render_image_hip(1, tile_counts=counts, stats=st)
    single = scene.render_image_hip(1, split_long_tiles=False)
    assert torch.equal(split, single)
    d = st["n_instances"]
    threshold = max(1024, 4 * (d // (ntx * nty)))
    n_long = int((counts > threshold).sum().item())
    print("clustered 150k: D = %d, mean list %.0f, longest %d, threshold %d, %d long tiles" % (
        d, d / (ntx * nty), int(counts.max().item()), threshold, n_long))
    assert 0 < n_long < 512 and int(counts.max().item()) > 8 * d / (ntx * nty)
    a = scene.render_image_hip(1, layout="hw3")
    b = scene.render_image_hip(1, layout="hw3", split_long_tiles=False)
    assert torch.equal(a, b) and torch.equal(a.permute(1, 0, 2), split)
    pre, ref, inst = _oracle_frame(scene, sc)
    assert inst == d
    assert np.max(np.abs(split.cpu().numpy() - ref)) <= PIXEL_TOL
    # a tile window and a captured frame go through the same split
    part = scene.render_image_hip(1, tile_window=(10, 30, 2, 20))
    assert torch.equal(part[160:480, 32:320], split[160:480, 32:320])
    frame = scene.capture_frame(1)
    frame.out.fill_(3.0)
    frame.replay()
    assert torch.equal(frame.confirm(), split)
    # more long tiles than helper slots: the rest stay on one wave, the frame is still the same
    sc2 = make_scene(400_000, 1280, 800, seed=5, cluster_fraction=0.9, cluster_area=0.4, sigma_ln=0.2)
    scene2 = _scene_from_arrays(tmp_path / "b", sc2)
    ntx2, nty2 = strips.tiles_along(1280, 16), strips.tiles_along(800, 16)
    counts2 = torch.zeros(ntx2 * nty2, dtype=torch.int32, device="cuda:0")
    st2 = {}
    s2 = scene2.render_image_hip(1, tile_counts=counts2, stats=st2)
    thr2 = max(1024, 4 * (st2["n_instances"] // (ntx2 * nty2)))
    print("many long tiles: %d above %d" % (int((counts2 > thr2).sum().item()), thr2))
    assert torch.equal(s2, scene2.render_image_hip(1, split_long_tiles=False))


def test_tiles_handed_out_by_list_length_render_the_same_frame(tmp_path):
    """GSX_FLAG_TILE_SCHEDULE: one more kernel ranks the tiles by list length and the compositing launch hands
    them to the SIMDs in that order (default from 300 000 Gaussians up).  Which workgroup composites which tile
    must not touch a pixel: both rule sets, both layouts, a tile window, a captured frame, a frame with long
    tiles and an empty frame equal the unscheduled frame bit for bit -- i.e. the schedule is a permutation of
    the tiles whatever their lengths (a tile missed or taken twice would show)."""
    _need_gpu()
    from intro_to_gaussian_splatting_amd.synthetic import make_scene

    w, h = 640, 400
    sc = make_scene(60_000, w, h, seed=11, cluster_fraction=0.5, cluster_area=0.05, sigma_ln=1.0, behind_fraction=0.1)
    scene = _scene_from_arrays(tmp_path, sc)
    for semantics in ("ref_cpu", "std_3dgs"):
        for layout in ("wh3", "hw3"):
            plain = scene.render_image_hip(1, layout=layout, semantics=semantics, tile_schedule=False)
            ranked = scene.render_image_hip(1, layout=layout, semantics=semantics, tile_schedule=True)
            assert plain.abs().sum() > 0 and torch.equal(plain, ranked), (semantics, layout)
    plain = scene.render_image_hip(1, tile_schedule=False)
    part = scene.render_image_hip(1, tile_window=(3, 29, 1, 17), tile_schedule=True)
    assert torch.equal(part[48:464, 16:272], plain[48:464, 16:272])
    one_wave = scene.render_image_hip(1, tile_schedule=True, split_long_tiles=False)
    assert torch.equal(one_wave, plain)
    # a window of a single tile, and tile counts that are no multiple of anything
    tiny = scene.render_image_hip(1, tile_window=(20, 21, 12, 13), tile_schedule=True)
    assert torch.equal(tiny[320:336, 192:208], plain[320:336, 192:208])
    # nothing in view: every list is empty, the schedule is still a permutation and the frame is zero
    (tmp_path / "behind").mkdir()
    behind = make_scene(5_000, w, h, seed=12, behind_fraction=1.0)
    st = {}
    empty = _scene_from_arrays(tmp_path / "behind", behind).render_image_hip(1, tile_schedule=True, stats=st)
    assert st["n_instances"] == 0 and float(empty.abs().sum()) == 0.0


@pytest.mark.parametrize("n,mode,lds_cap,kind", [
    (16_384, 1, 0, "random"), (20_001, 1, 0, "random"), (100_003, 1, 0, "depth"), (131_072, 1, 0, "random"),
    (131_073, 1, 0, "depth"), (1_000_000, 1, 0, "depth"), (1_000_000, 0, 0, "depth"), (2_200_000, 1, 0, "random"),
    (300_000, 1, 0, "equal"), (300_000, 1, 0, "two"), (300_000, 1, 0, "sorted"), (300_000, 1, 0, "reverse"),
    (300_000, 1, 0, "adversarial"), (60_000, 1, 64, "depth"), (400_000, 1, 1024, "depth"), (400_000, 1, 1000, "two"),
    (50_000, 0, 0, "random"), (3_000, 0, 0, "depth"),
    (1, 1, 0, "random"), (63, 1, 0, "random"), (2_000, 1, 0, "depth"), (9_999, 1, 0, "equal"), (16_384, 1, 0, "two"),
    (16_000, 1, 0, "random"),
    # 1024 buckets (round 3; measured slower than the LSD passes, so no frame takes it -- the templates it shares with
    # the 256-bucket route are held to the same bar): sizes around and far beyond 2^20, ties, one value, adversarial
    # sample positions, a shrunken LDS capacity, few keys in many buckets
    (2_200_000, 2, 0, "depth"), (5_000_000, 2, 0, "depth"), (5_000_000, 2, 0, "random"), (2_200_000, 2, 0, "adversarial"),
    (5_000_000, 2, 0, "adversarial"), (2_200_000, 2, 2048, "depth"), (1_200_000, 2, 0, "two"), (1_200_000, 2, 0, "equal"),
    (70_000, 2, 0, "depth"), (2_200_000, 2, 0, "sorted"), (2_200_000, 2, 0, "reverse"), (5_000_000, 2, 0, "strip"),
    # the route gsx_render_forward takes (mode -1): by n alone, and with a hint of how many keys are kept
    (1_200_000, -1, 0, "depth"), (2_200_000, -1, 0, "depth"), (5_000_000, -1, 0, "strip"), (5_000_000, -1, 0, "depth"),
    (7_000_000, -1, 0, "depth"),
    # mode 4: the LSD passes carrying the rectangles along, packed into 4 bytes (what a frame of more than 1.5M kept
    # Gaussians on a tile grid of up to 256 x 256 takes); below 2^20 keys it is the plain LSD route
    (5_000_000, 4, 0, "depth"), (2_200_000, 4, 0, "random"), (1_048_576, 4, 0, "two"), (3_000_001, 4, 0, "strip"),
    (600_000, 4, 0, "depth")])
def test_depth_sort_paths_are_the_stable_argsort(n, mode, lds_cap, kind):
    """The depth sort of the whole-path entry (gsx_debug_depth_sort in libgsx_test.so): four compacting LSD passes
    (mode 0) and the sample-partitioned sorts (mode 1: 2048 / 8192 samples -> 255 splitters, mode 2: 8192 samples ->
    1023 splitters; one partition pass, one in-LDS sort per bucket); mode -1 = whatever depth_sort_route picks for
    (n, kept_hint).  All must return the STABLE argsort of the keys that are kept (< 0xFFFFFFFE), the rectangles
    gathered into rank order, and the kept / culled counts.  lds_cap > 0 shrinks the bucket kernel's LDS capacity
    so that buckets go through its global-memory path; "adversarial" puts all small keys on the sampled positions,
    so that one bucket receives almost everything; "strip" drops 7 of 8 keys, like a rank that owns an eighth of
    the frame, and hands the sort the matching hint."""
    _need_gpu()
    import ctypes

    from intro_to_gaussian_splatting_amd import _ffi

    lib = _ffi.load_test_hooks()
    fn = lib.gsx_debug_depth_sort
    rs = np.random.RandomState(n % 9973 + (mode & 3))
    if kind == "random":
        keys = rs.randint(0, 2 ** 32 - 2, size=n, dtype=np.uint64).astype(np.uint32)
        keys[rs.uniform(size=n) < 0.1] = rs.randint(0, 50, size=int((rs.uniform(size=n) < 0.1).sum()) or 1)[0]   # duplicates
    elif kind in ("depth", "strip"):
        keys = rs.uniform(0.2, 40.0, n).astype(np.float32).view(np.uint32).copy()
        keys[::7] = keys[3]                                              # ties: index order decides
    elif kind == "equal":
        keys = np.full(n, 0x40490FDB, np.uint32)
    elif kind == "two":
        keys = np.where(rs.uniform(size=n) < 0.5, 0x3F800000, 0x3F800001).astype(np.uint32)
    elif kind in ("sorted", "reverse"):
        keys = np.sort(rs.uniform(0.2, 40.0, n).astype(np.float32)).view(np.uint32).copy()
        keys = keys[::-1].copy() if kind == "reverse" else keys
    elif n < 2048:
        keys = rs.randint(0, 2 ** 31, size=n).astype(np.uint32)
    else:   # adversarial: the sampled positions (both sample sizes) hold tiny keys, everything else is large and distinct-ish
        keys = (0x41000000 + rs.randint(0, 2 ** 22, size=n)).astype(np.uint32)
        for ns in (2048, 8192):
            keys[(np.arange(ns, dtype=np.uint64) * n // ns).astype(np.int64)] = rs.randint(1, 1000, ns)
    drop = rs.uniform(size=n)
    if kind == "strip":
        keys[drop < 0.875] = 0xFFFFFFFE
        keys[drop < 0.05] = 0xFFFFFFFF
    else:
        keys[drop < 0.08] = 0xFFFFFFFF
        keys[(drop >= 0.08) & (drop < 0.2)] = 0xFFFFFFFE
    kept = np.nonzero(keys < 0xFFFFFFFE)[0]
    expect = kept[np.argsort(keys[kept], kind="stable")]
    rect = rs.randint(0, 256 if mode == 4 else 65535, size=(n, 4)).astype(np.uint16)
    d_keys = torch.from_numpy(keys.view(np.int32).copy()).cuda()
    d_rect = torch.from_numpy(rect.view(np.int16).copy()).cuda()
    d_rrect = torch.zeros_like(d_rect)
    d_order = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    nbytes = 24 * n + 8192 + lib.gsx_workspace_bytes(n, 16, 16, 16, 1)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device="cuda:0")
    counts = (ctypes.c_int64 * 3)()
    hint = int(kept.size) if kind == "strip" else 0
    p = lambda t: t.data_ptr()   # noqa: E731
    rc = fn(p(d_keys), n, p(d_rect), p(d_rrect), p(d_order), mode, lds_cap, hint, counts, p(scratch), nbytes,
            torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    # (modes the hook does not know, and a scratch that is too small, are refused -- csrc/gsx_debug.h)
    assert fn(p(d_keys), n, p(d_rect), p(d_rrect), p(d_order), 3, lds_cap, hint, counts, p(scratch), nbytes,
              torch.cuda.current_stream().cuda_stream) == _ffi.GSX_ERR_INVALID_ARGUMENT
    assert fn(p(d_keys), n, p(d_rect), p(d_rrect), p(d_order), mode, lds_cap, hint, counts, p(scratch), 4 * n,
              torch.cuda.current_stream().cuda_stream) == _ffi.GSX_ERR_WORKSPACE_TOO_SMALL
    assert counts[0] == kept.size and counts[1] == int((keys == 0xFFFFFFFF).sum())
    if mode == -1:      # routes: 0 LSD, 1 = 256 buckets, (2 = 1024 buckets: no frame takes it), 3 = one workgroup
        want = {1_200_000: 1, 2_200_000: 0, 7_000_000: 0}.get(n, 1 if kind == "strip" else 0)
        assert counts[2] == want, (counts[2], want)
    got = d_order[:kept.size].cpu().numpy().astype(np.int64)
    assert np.array_equal(got, expect)
    assert np.array_equal(d_rrect[:kept.size].cpu().numpy().view(np.uint16), rect[expect])


def test_skipped_records_stay_below_tolerance_however_long_the_list(tmp_path):
    """Round-2 verdict, weak #2: the compositing kernels do not stage a record whose alpha stays below a threshold
    on the whole tile.  With a flat 2^-26 threshold 6 711 such records could add up to the 1e-4 tolerance; every
    skipped record is now CHARGED -- 2^-26, 2^-33 or 2^-40 by the class of its bound -- and a batch skips only the
    classes that still fit a budget of 2^-17 = 7.6e-6 of colour (gsx_blend.hip: kSkipBudget, stage_records; since
    round 4 the test and the budget are per 8x8 BLOCK of the tile, whose 16 lanes walk a list of their own).  16 384 thin Gaussians whose peak alpha on the middle tile lies between 2^-26.05 and 2^-27 -- a flat
    threshold would skip every one and lose ~1.5e-4 of colour there -- must match the C restatement to 1e-5 on that
    tile and to 1e-4 everywhere.  Stage-2 entry point (hand-made stage-1 arrays, splat/c/render.cu:90-101 argument
    list), both kernel families."""
    _need_gpu()
    from intro_to_gaussian_splatting_amd import render_preprocessed
    from oracle import c_oracle, cpu_ref

    rs = np.random.RandomState(5)
    n, w, h, tile = 16_384, 80, 80, 16                  # tiles (0..3) x (0..3); the tile under test: x 16..31, y 16..31
    sx, sy = 2.0, 20.0                                   # thin along x, long along y: radius = ceil(3 * 20) = 60
    op = 1.0 / (1.0 + np.exp(-1.0 / (1.0 + np.exp(-10.0))))          # sigmoid(sigmoid(10)) = 0.7311
    # alpha at the tile's nearest column (x = 31): op * exp(-dx^2 / (2 sx^2)) = 2^-e  ->  dx
    e = rs.uniform(26.05, 27.0, n)
    dx = sx * np.sqrt(2.0 * (e * np.log(2.0) + np.log(op)))
    means = np.stack([31.0 + dx, rs.uniform(20.0, 28.0, n)], axis=1).astype(np.float32)
    inv = np.zeros((n, 2, 2), np.float32)
    inv[:, 0, 0], inv[:, 1, 1] = 1.0 / sx ** 2, 1.0 / sy ** 2
    r = np.float32(60.0)
    pre = cpu_ref.Preprocessed(
        points=means, colors=np.ones((n, 3), np.float32), covariance_2d=np.zeros((n, 2, 2), np.float32),
        depths=np.arange(n, dtype=np.float32), inverse_covariance_2d=inv, radius=np.full(n, r, np.float32),
        points_xy=means, min_x=np.floor(means[:, 0] - r), min_y=np.floor(means[:, 1] - r),
        max_x=np.ceil(means[:, 0] + r), max_y=np.ceil(means[:, 1] + r),
        sigmoid_opacity=np.full((n, 1), 1.0 / (1.0 + np.exp(-10.0)), np.float32), order=np.arange(n))
    ref, _, inst = c_oracle.render(pre, w, h, tile)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    args = [t(pre.points), t(pre.colors), t(pre.inverse_covariance_2d), t(pre.min_x), t(pre.max_x), t(pre.min_y),
            t(pre.max_y), t(pre.sigmoid_opacity)]
    st = {}
    img = render_preprocessed(h, w, tile, *args, stats=st).cpu().numpy()
    assert st["n_instances"] == inst == 16 * n                       # every Gaussian is listed for every tile
    col = ref[31, 16:32, 0]
    print("skip test: reference colour at the tile's nearest column %.3g .. %.3g (what a flat 2^-26 skip would lose)" % (
        float(col.min()), float(col.max())))
    assert col.min() > 1.2e-4                                        # skipping them all WOULD break the tolerance
    assert np.max(np.abs(img[16:32, 16:32] - ref[16:32, 16:32])) <= 1e-5
    assert np.max(np.abs(img - ref)) <= PIXEL_TOL
    # and records just below the length-aware threshold (2^-31 at this list length) may be skipped: their sum
    # stays far below the tolerance either way
    e2 = rs.uniform(31.2, 33.0, n)
    dx2 = sx * np.sqrt(2.0 * (e2 * np.log(2.0) + np.log(op)))
    means2 = np.stack([31.0 + dx2, rs.uniform(20.0, 28.0, n)], axis=1).astype(np.float32)
    pre2 = pre._replace(points=means2, points_xy=means2, min_x=np.floor(means2[:, 0] - r), max_x=np.ceil(means2[:, 0] + r))
    ref2, _, _ = c_oracle.render(pre2, w, h, tile)
    args2 = [t(pre2.points)] + args[1:3] + [t(pre2.min_x), t(pre2.max_x)] + args[5:]
    img2 = render_preprocessed(h, w, tile, *args2).cpu().numpy()
    assert np.max(np.abs(img2 - ref2)) <= 1e-5


def _report_against_port(tag, img, port, exact):
    d_port = np.abs(img - port).max(axis=2)
    d_exact = np.abs(img - exact).max(axis=2)
    port_err = np.abs(port.astype(np.float64) - exact).max(axis=2)
    hist = np.histogram(d_port, bins=[0.0, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, np.inf])[0]
    print("%s: kernel vs port %.3g (per pixel <=1e-7 / 1e-6 / 1e-5 / 1e-4 / 1e-3 / above: %s); for information, against the "
          "same rules in float64: kernel %.3g, port %.3g" % (tag, d_port.max(), list(hist), d_exact.max(), port_err.max()))
    return d_port


def test_clustered_1m_scene_within_tolerance_of_the_port(tmp_path):
    """The heavy-tailed stress scene of bench.py (--workload c3_clustered: 1M Gaussians, half of them inside 5 % of
    a 1080p frame, footprint sigma_ln 1.0; longest tile list ~20x the mean).  Not a BASELINE config.  Long, thin,
    rotated footprints seen far along their ridge are where the REFERENCE's float32 grouping of d Q d^T loses up to
    1e-3 of alpha -- and since the restatement executes the reference's operations in the reference's order (pinned on
    such footprints by tests/golden/needle_160x160_n110: 6e-8), that loss is part of the result to be matched: the
    kernel keeps the reference's own evaluation on every ill-conditioned record (kKindRefOrder, gsx_blend.hip).
    The bar is the plain one: counts identical to the port, EVERY pixel within 1e-4 of it."""
    _need_gpu()
    from intro_to_gaussian_splatting_amd.synthetic import make_scene
    from oracle import c_oracle

    w, h = 1920, 1080
    sc = make_scene(1_000_000, w, h, seed=0, cluster_fraction=0.5, cluster_area=0.05, sigma_ln=1.0)
    scene = _scene_from_arrays(tmp_path, sc)
    st = {}
    img = scene.render_image_hip(1, stats=st).cpu().numpy().astype(np.float64)
    cam = _oracle_cam(scene)
    pre = c_oracle.preprocess(sc["points"], scene.gaussians.colors.cpu().numpy(), sc["scales"], sc["quaternions"],
                              sc["opacity"], cam)
    port, _, inst = c_oracle.render(pre, w, h, 16)
    exact, _, _ = c_oracle.render(pre, w, h, 16, exact=True)
    assert st["n_visible"] == pre.points.shape[0] and st["n_instances"] == inst
    d_port = _report_against_port("clustered 1M", img, port, exact)
    assert d_port.max() <= PIXEL_TOL


def test_the_long_tiles_of_a_view_are_a_fixed_point(tmp_path):
    """Which tiles are composited by four helper waves is decided by what they cost in the previous frame
    (GsxParams.hints, tile_ranges_kernel).  Four quarters and one wave do not report the same cost, and round 6 found nine
    tiles of this scene within that difference of the threshold: long, not long, long ... -- the view's frames alternated
    between 0.476 and 0.556 ms (tools/frame_sequence.py).  A tile that was long now stays long down to 75 % of the
    threshold (LongTiles::stay_pct): replayed frames of one view must settle on ONE set of long tiles -- and be the same
    frame bit for bit whoever is long."""
    _need_gpu()
    from intro_to_gaussian_splatting_amd.synthetic import make_scene

    w, h = 1920, 1080
    sc = make_scene(1_000_000, w, h, seed=0, cluster_fraction=0.5, cluster_area=0.05, sigma_ln=1.0)
    scene = _scene_from_arrays(tmp_path, sc)
    first = scene.render_image_hip(1, use_hints=False).clone()
    frame = scene.capture_frame(1)
    ntiles = ((w + 15) // 16) * ((h + 15) // 16)
    lens_at = (64 + 256 + 2048) * 4          # csrc/gsx_plan.h: hints_layout (header, splitters, samples, then the tiles' costs)
    long_sets = []
    for rep in range(12):
        frame.replay()
        assert torch.equal(frame.confirm(), first), rep
        cost = frame._hints[lens_at:lens_at + 4 * ntiles].cpu().numpy().view(np.uint32)
        long_sets.append(np.nonzero(cost >> 31)[0])
    print("long tiles per replay:", [int(s.size) for s in long_sets])
    assert long_sets[-1].size >= 16                  # (the scene has them: ~180 of 8 160 tiles)
    for s in long_sets[-4:]:
        assert np.array_equal(s, long_sets[-1])


def test_trained_like_1m_scene_within_tolerance_of_the_port(tmp_path):
    """BASELINE config 3 AS WRITTEN is a trained Treehill .ply -- needle footprints, heavy-tailed sizes -- and it is
    not available offline.  synthetic.make_trained_like_scene generates the nearest thing: 1M Gaussians in 24 clusters
    + floaters, axis ratios to 50:1, log-normal sizes with sigma_ln 1.2, bimodal opacity, degree-3 spherical harmonics.
    Same bar as everywhere: counts identical to the port, every pixel within 1e-4 of it (bench.py --workload
    c3_trainedlike reports the same as parity_ok)."""
    _need_gpu()
    from intro_to_gaussian_splatting_amd import GaussianScene, Gaussians
    from intro_to_gaussian_splatting_amd.synthetic import make_trained_like_scene, write_colmap_text
    from oracle import c_oracle

    w, h = 1920, 1080
    sc = make_trained_like_scene(1_000_000, w, h, seed=0)
    write_colmap_text(str(tmp_path), sc)
    g = Gaussians.from_arrays(sc["points"], sc["colors_0_255"], sc["scales"], sc["quaternions"], sc["opacity"], device="cuda:0")
    g.sh, g.sh_degree = torch.from_numpy(sc["sh"]).cuda().contiguous(), 3
    scene = GaussianScene(str(tmp_path), g)
    st = {}
    img = scene.render_image_hip(1, stats=st).cpu().numpy().astype(np.float64)
    colors = scene._colors(1).cpu().numpy()             # the view-dependent colours of this camera (gsx_sh_to_rgb)
    pre = c_oracle.preprocess(sc["points"], colors, sc["scales"], sc["quaternions"], sc["opacity"], _oracle_cam(scene))
    port, _, inst = c_oracle.render(pre, w, h, 16)
    exact, _, _ = c_oracle.render(pre, w, h, 16, exact=True)
    assert st["n_visible"] == pre.points.shape[0] and st["n_instances"] == inst
    d_port = _report_against_port("trained-like 1M (D = %d)" % inst, img, port, exact)
    assert d_port.max() <= PIXEL_TOL


def test_needles_through_every_kernel_family(tmp_path):
    """tests/golden/needle_160x160_n110 (250:1 footprints, rendered by the reference itself; the same rules in float64
    are 5.4e-4 away from it): the tile-16 kernel (whole path and stage-2 entry), the any-tile-size kernels, the
    long-tile quarter kernel (every tile on four waves) and another tile size all land within 1e-5 of the reference's
    image -- the ill-conditioned records take the reference's own operations in each of them."""
    _need_gpu()
    from oracle import c_oracle

    g = load_golden("needle_160x160_n110")
    scene = _scene_from_golden(tmp_path, g)
    ref = g["image"]
    a = scene.render_image_hip(1).cpu().numpy()
    b = scene.render_image_hip(1, generic_kernels=True).cpu().numpy()
    c = scene.render_image_hip(1, split_long_tiles=False).cpu().numpy()
    for name, img in (("tile16", a), ("generic", b), ("no split", c)):
        assert np.max(np.abs(img - ref)) <= 1e-5, (name, np.max(np.abs(img - ref)))
    assert np.array_equal(a, b) and np.array_equal(a, c)
    sc = {k: g[k] for k in ("points", "scales", "quaternions", "opacity")}
    for tile in (8, 32):
        _, port, inst = _oracle_frame(scene, sc, tile)
        st = {}
        img = scene.render_image_hip(1, tile_size=tile, stats=st).cpu().numpy()
        assert st["n_instances"] == inst and np.max(np.abs(img - port)) <= 1e-5
    # the long-tile quarter kernel: the heavy-tailed scene of test_long_tiles_on_four_waves_... with every 64th Gaussian
    # stretched 40-fold into a needle -- tiles composited by four waves hold reference-order records; same frame as one
    # wave per tile bit for bit, within tolerance of the port
    from intro_to_gaussian_splatting_amd import strips
    from intro_to_gaussian_splatting_amd.synthetic import make_scene

    w, h = 640, 400
    sc2 = make_scene(150_000, w, h, seed=4, cluster_fraction=0.5, cluster_area=0.05, sigma_ln=1.0)
    sc2["scales"] = sc2["scales"].copy()
    sc2["scales"][::64, 0] *= 40.0
    scene2 = _scene_from_arrays(tmp_path / "long", sc2)
    counts = torch.zeros(strips.tiles_along(w, 16) * strips.tiles_along(h, 16), dtype=torch.int32, device="cuda:0")
    st = {}
    split = scene2.render_image_hip(1, tile_counts=counts, stats=st)
    assert int((counts > max(1024, 4 * (st["n_instances"] // counts.numel()))).sum().item()) > 0      # long tiles exist
    assert torch.equal(split, scene2.render_image_hip(1, split_long_tiles=False))
    assert torch.equal(split, scene2.render_image_hip(1, generic_kernels=True))
    pre2, port2, inst2 = _oracle_frame(scene2, sc2)
    exact2, _, _ = c_oracle.render(pre2, w, h, 16, exact=True)
    assert inst2 == st["n_instances"]
    d = _report_against_port("needles in long tiles", split.cpu().numpy().astype(np.float64), port2, exact2)
    assert d.max() <= PIXEL_TOL


def test_conic_whose_float32_determinant_cancelled_takes_the_reference_order(tmp_path):
    """Found by tools/fuzz.py (big profile, seed 301): a 290:1 needle 2 000 px long whose 2D covariance determinant cancels
    in float32 -- the reference inverts it all the same (utils.py:383 floors the determinant), and the conic that comes
    out, (16.61, -24.63; -24.63, 36.51), has c = |Q01 + Q10| / 2 sqrt(Q00 Q11) = 1 - 6.6e-8: in float32 exactly 1.0.  The
    flag test of pack_record excluded c >= 1 and composited the MOST ill-conditioned record by the completed square:
    0.058 off the reference's operations.  Through the stage-2 entry (the record as the reference's stage 1 made it),
    alone and among regular footprints, on every kernel family."""
    _need_gpu()
    from intro_to_gaussian_splatting_amd import render_preprocessed
    from oracle import c_oracle, cpu_ref

    w, h = 1456, 1139
    rs = np.random.RandomState(5)
    m = 41
    pts = np.stack([rs.uniform(600, 900, m), rs.uniform(550, 850, m)], axis=1).astype(np.float32)
    inv = np.zeros((m, 2, 2), np.float32)
    sig = rs.uniform(6.0, 30.0, m).astype(np.float32)
    inv[:, 0, 0] = inv[:, 1, 1] = 1.0 / (sig * sig)
    rad = np.ceil(3.0 * sig).astype(np.float32)
    op = rs.uniform(0.05, 0.9, m).astype(np.float32)
    k = 20       # the needle, in the middle of the depth order
    pts[k] = (755.3137817382812, 699.432861328125)
    inv[k] = [[16.614168167114258, -24.628461837768555], [-24.628463745117188, 36.50867462158203]]
    rad[k] = 1980.0
    op[k] = 0.011413033120334148
    pre = cpu_ref.Preprocessed(points=pts, colors=rs.uniform(0.1, 1.0, (m, 3)).astype(np.float32), covariance_2d=np.zeros((m, 2, 2), np.float32),
                               depths=np.arange(m, dtype=np.float32) + 1.0, inverse_covariance_2d=inv, radius=rad, points_xy=pts,
                               min_x=pts[:, 0] - rad, min_y=pts[:, 1] - rad, max_x=pts[:, 0] + rad, max_y=pts[:, 1] + rad,
                               sigmoid_opacity=op.reshape(-1, 1), order=np.arange(m))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    for name, sub in (("alone", pre._replace(**{f: np.asarray(getattr(pre, f))[k:k + 1] for f in pre._fields})), ("among others", pre)):
        for tile in (16, 8):
            ref, _, inst = c_oracle.render(sub, w, h, tile)
            st = {}
            img = render_preprocessed(h, w, tile, t(sub.points), t(sub.colors), t(sub.inverse_covariance_2d), t(sub.min_x), t(sub.max_x),
                                      t(sub.min_y), t(sub.max_y), t(sub.sigmoid_opacity), stats=st).cpu().numpy()
            assert st["n_instances"] == inst
            d = float(np.abs(img - ref).max())
            assert d <= PIXEL_TOL, (name, tile, d)
        assert float(ref.max()) > 1e-3, name        # (the needle is in the picture)


@pytest.mark.parametrize("seed,big", [(30, False), (7, False), (2, True)])
def test_corners_of_the_parameter_space(tmp_path, seed, big):
    """tools/fuzz.py's `extreme` profile: needles to 3000:1, pancakes, specks of a hundredth of a pixel, opacity logits of
    +-12, many centres on one spot.  Seeds 30 and (big) 2 were 0.56 / 0.80 off: a needle seen from 1 000 px has the
    reference's own exponent off by more than one -- its alpha is the exact one times e^(+-delta), not plus a small
    delta alpha --, and a flagged record was demoted / dropped from a block's list by the EXACT alpha there."""
    _need_gpu()
    from tools.fuzz_scene import fuzz_scene

    _, sc, w, h, tile, n, _ = fuzz_scene(seed, big, True)
    scene = _scene_from_arrays(tmp_path, sc)
    _, ref, inst = _oracle_frame(scene, sc, tile)
    st = {}
    img = scene.render_image_hip(1, tile_size=tile, stats=st)
    assert st["n_instances"] == inst
    d = float(np.abs(img.cpu().numpy() - ref).max())
    assert d <= PIXEL_TOL, (seed, big, d)
    assert torch.equal(img, scene.render_image_hip(1, tile_size=tile, generic_kernels=True))
    assert torch.equal(img, scene.render_image_hip(1, tile_size=tile, split_long_tiles=False))


def test_plain_footprints_instance_only_where_no_tile_needs_the_other(tmp_path, monkeypatch):
    """GsxFrameStats.n_redo / GSX_FLAG_PLAIN_FOOTPRINTS: a view without ill-conditioned footprints reports n_redo = 0, its
    next frame runs the compositing instance that cannot evaluate them (here also on a small window: the wrapper's size
    threshold is lowered) and is the same frame; a view with needles reports n_redo > 0 and keeps the other instance; a
    frame that wrongly took the plain one reports the tiles it left undone and the wrapper renders it again -- on the
    synchronising path at once, for enqueued frames in confirm_frames(), for a captured frame by raising in confirm()."""
    _need_gpu()
    from intro_to_gaussian_splatting_amd import _ffi
    from intro_to_gaussian_splatting_amd.synthetic import make_scene

    from intro_to_gaussian_splatting_amd import gaussian_scene as wrapper

    w, h = 320, 192
    sc = make_scene(20_000, w, h, seed=9)
    scene = _scene_from_arrays(tmp_path / "plain", sc)
    st = {}
    a = scene.render_image_hip(1, stats=st).clone()
    assert st["n_redo"] == 0
    key = (1, 16, None, "ref_cpu")
    assert scene._n_redo_seen[key] == 0
    assert not scene.capture_frame(1)._plain_footprints        # (240 tiles: the plain instance pays from 16 384)
    monkeypatch.setattr(wrapper, "_PLAIN_MIN_TILES", 1)
    st = {}
    assert torch.equal(scene.render_image_hip(1, stats=st), a) and st["n_redo"] == 0        # (issued with GSX_FLAG_PLAIN_FOOTPRINTS)
    frame = scene.capture_frame(1, headroom=8.0)        # (room for the pairs of the needles grown below)
    assert frame._plain_footprints
    frame.replay()
    assert torch.equal(frame.confirm(), a)
    # needles: counted, and the view keeps the instance that evaluates them
    sc2 = dict(sc)
    sc2["scales"] = sc["scales"].copy()
    sc2["scales"][::8, 0] *= 80.0
    needles = _scene_from_arrays(tmp_path / "needles", sc2)
    st = {}
    b = needles.render_image_hip(1, stats=st).clone()
    assert st["n_redo"] > 0 and needles._n_redo_seen[key] > 0
    _, port, inst = _oracle_frame(needles, sc2)
    assert st["n_instances"] == inst and np.max(np.abs(b.cpu().numpy() - port)) <= PIXEL_TOL
    assert not needles.capture_frame(1)._plain_footprints
    # a frame that wrongly takes the plain instance: the synchronising path notices and renders again ...
    needles._n_redo_seen[key] = 0
    st = {}
    assert torch.equal(needles.render_image_hip(1, stats=st), b) and st["n_redo"] > 0
    # ... an enqueued frame is redone by confirm_frames() ...
    needles._n_redo_seen[key] = 0
    out = torch.empty_like(b)
    needles.render_image_hip(1, out=out, no_sync=True)
    assert needles.confirm_frames() == 1 and torch.equal(out, b)
    # ... and a frame captured for a scene that has since grown needles says so
    scene.gaussians.scales[::8, 0] *= 80.0
    frame.replay()
    with pytest.raises(_ffi.GsxError, match="GSX_FLAG_PLAIN_FOOTPRINTS"):
        frame.confirm()


def test_both_compositing_instances_render_the_same_frames(tmp_path, monkeypatch):
    """The compositing launch has two instances (gsx_blend.hip: REF): the one every frame runs and the plain one of
    GSX_FLAG_PLAIN_FOOTPRINTS, which the wrapper takes from 16 384 tiles.  With that threshold lowered the second frame
    of a view without ill-conditioned footprints runs the plain instance: same frame bit for bit -- on reference-made
    fixtures (whose images both are held against), on a heavy-tailed scene with long tiles on four waves, on a tile
    window, in both layouts and through a captured frame."""
    _need_gpu()
    from intro_to_gaussian_splatting_amd import gaussian_scene as wrapper
    from intro_to_gaussian_splatting_amd.synthetic import make_scene

    cases = [(name, _scene_from_golden(tmp_path / name, load_golden(name)), load_golden(name))
             for name in ("small_64x48_n300", "c1_256x256_n2000", "dense_48x48_n1500", "defaults_64x64_n800")]
    sc = make_scene(150_000, 640, 400, seed=4, cluster_fraction=0.5, cluster_area=0.05, sigma_ln=1.0)
    sc["scales"] = np.repeat(sc["scales"][:, :1], 3, axis=1)      # round footprints: heavy-tailed sizes, no needles
    cases.append(("heavy-tailed", _scene_from_arrays(tmp_path / "heavy", sc), None))
    first = {}
    for name, scene, g in cases:
        st = {}
        tile = int(g["tile"]) if g is not None else 16
        first[name] = scene.render_image_hip(1, tile_size=tile, stats=st).clone()
        assert st["n_redo"] == 0 and not st["plain_footprints"], (name, st)
    monkeypatch.setattr(wrapper, "_PLAIN_MIN_TILES", 1)
    for name, scene, g in cases:
        tile = int(g["tile"]) if g is not None else 16
        st = {}
        b = scene.render_image_hip(1, tile_size=tile, stats=st)
        assert st["plain_footprints"] == (tile == 16) and st["n_redo"] == 0, (name, st)
        assert torch.equal(b, first[name]), name
        if g is not None:
            assert np.max(np.abs(b.cpu().numpy() - g["image"])) <= PIXEL_TOL
        if tile == 16:
            st = {}
            hw = scene.render_image_hip(1, layout="hw3", stats=st)
            assert torch.equal(hw.permute(1, 0, 2), first[name]), name
    scene = cases[-1][1]
    st = {}
    part = scene.render_image_hip(1, tile_window=(10, 30, 2, 20), stats=st)        # (first frame of that window: the other instance)
    st = {}
    again = scene.render_image_hip(1, tile_window=(10, 30, 2, 20), stats=st)
    assert st["plain_footprints"] and torch.equal(again, part)
    assert torch.equal(part[160:480, 32:320], first["heavy-tailed"][160:480, 32:320])
    frame = scene.capture_frame(1)
    assert frame._plain_footprints
    frame.replay()
    assert torch.equal(frame.confirm(), first["heavy-tailed"])


def test_orbit_of_eight_poses_through_one_captured_frame(tmp_path):
    """What bench.py --camera-path times, at a size the oracle renders whole: eight cameras 1 degree apart on an orbit
    (synthetic.orbit_poses), ONE frame captured with a movable camera and re-aimed before every replay.  Every replay
    -- which finds the hints (splitters, tile costs, schedule, kept count) the PREVIOUS pose left -- equals the plain
    render of that camera bit for bit and fits the pair capacity of the graph; three of the poses are held against the
    C restatement (counts equal, pixels within 1e-4)."""
    _need_gpu()
    from intro_to_gaussian_splatting_amd import GaussianScene, Gaussians
    from intro_to_gaussian_splatting_amd.synthetic import make_scene, orbit_poses, write_colmap_text
    from oracle import c_oracle

    w, h = 640, 368
    sc = make_scene(60_000, w, h, seed=17)
    write_colmap_text(str(tmp_path), sc, extra_poses=orbit_poses(8))
    g = Gaussians.from_arrays(sc["points"], sc["colors_0_255"], sc["scales"], sc["quaternions"], sc["opacity"], device="cuda:0")
    scene = GaussianScene(str(tmp_path), g)
    ids = list(range(2, 10))
    assert sorted(scene.images) == [1] + ids
    refs = {i: scene.render_image_hip(i, use_hints=False).clone() for i in ids}
    assert not torch.equal(refs[2], refs[3])
    frame = scene.capture_frame(ids[4], movable_camera=True, headroom=1.3)
    for i in ids + ids[::-1] + [ids[0], ids[7], ids[3]]:         # neighbours, then jumps of several degrees
        frame.set_camera(i)
        frame.replay()
        nvis, d, room = frame.counts()
        assert d <= room, (i, d, room)
        assert torch.equal(frame.out, refs[i]), i
    for i in (ids[0], ids[4], ids[7]):
        pre = c_oracle.preprocess(sc["points"], g.colors.cpu().numpy(), sc["scales"], sc["quaternions"], sc["opacity"],
                                  _oracle_cam(scene, i))
        img, _, inst = c_oracle.render(pre, w, h, 16)
        st = {}
        got = scene.render_image_hip(i, stats=st)
        assert st["n_visible"] == pre.points.shape[0] and st["n_instances"] == inst
        assert np.max(np.abs(got.cpu().numpy() - img)) <= PIXEL_TOL


def test_render_images_overlaps_the_copy_and_returns_the_same_frames(tmp_path):
    """GaussianScene.render_images: the reference's image-after-image loop with the device-to-host copy of frame i
    overlapped with the rendering of frame i + 1 (two device frames, two pinned host buffers): every yielded host
    tensor equals render_image of that camera bit for bit, also when the same camera repeats and when the sequence has
    one or no element."""
    _need_gpu()
    from intro_to_gaussian_splatting_amd import GaussianScene, Gaussians
    from intro_to_gaussian_splatting_amd.synthetic import make_scene, orbit_poses, write_colmap_text

    sc = make_scene(30_000, 480, 272, seed=23)
    write_colmap_text(str(tmp_path), sc, extra_poses=orbit_poses(4, step_deg=2.0))
    g = Gaussians.from_arrays(sc["points"], sc["colors_0_255"], sc["scales"], sc["quaternions"], sc["opacity"], device="cuda:0")
    scene = GaussianScene(str(tmp_path), g)
    order = [1, 2, 3, 3, 4, 5, 1, 2]
    refs = {i: scene.render_image(i).clone() for i in set(order)}
    got = [f.clone() for f in scene.render_images(order)]      # (a yielded tensor is reused two frames later: clone)
    assert len(got) == len(order)
    for i, f in zip(order, got):
        assert not f.is_cuda and torch.equal(f, refs[i]), i
    assert [f.clone() for f in scene.render_images([4])][0].equal(refs[4])
    assert list(scene.render_images([])) == []


@pytest.mark.parametrize("world,extra", [(2, []), (3, ["--balance"])])
def test_ranks_sharing_one_gpu_assemble_the_single_gpu_frame(world, extra):
    """bench.py's N > 1 path end to end on the ONE GPU of the test box (tools/bench_two_ranks_one_gpu.py: `world` processes
    share GPU 0, the process group is gloo -- RCCL refuses two ranks per device -- everything else is the real code):
    strips, the overlapped gather (every rank composites its strip in four parts behind HIP events the library records,
    sends part j from a communication stream while part j + 1 is composited, rank 0 posts its receive groups first),
    bench.py's self-check and timing protocol.  The frame rank 0 assembles must equal the single-GPU frame bit for bit
    (`strips_equal_single_gpu`) and the line must say that the overlapped path was the one that ran."""
    _need_gpu()
    import json
    import subprocess
    import sys

    tool = os.path.join(ROOT, "tools", "bench_two_ranks_one_gpu.py")
    r = subprocess.run([sys.executable, tool, str(world), "--workload", "c2", "--steps", "3", "--warmup", "1", "--repeats", "1",
                        "--no-cpu-baseline", "--streams", "1"] + extra, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == world and d["strips_equal_single_gpu"] is True
    assert "sub-strips sent while the next is composited" in d["config"]["parallelism"], d["config"]["parallelism"]


def test_compositing_in_parts_equals_the_one_launch_frame(tmp_path):
    """GsxParams.n_substrips (round 4): projection, depth order and binning once, the compositing launch once per part of
    the window, an event of the caller's behind each part.  Whole frames and a strip window, both layouts, 2 .. 16
    parts incl. empty ones, with hints and long tiles in play: every frame equals the one-launch frame bit for bit,
    and a stream that waits for event k alone sees part k complete.  Rule sets without a partial launch (std_3dgs
    here) composite in one launch and still mark every event."""
    _need_gpu()
    from intro_to_gaussian_splatting_amd import _hip, strips
    from intro_to_gaussian_splatting_amd.synthetic import make_scene

    w, h, tile = 1280, 720, 16
    sc = make_scene(150_000, w, h, seed=5, cluster_fraction=0.4, cluster_area=0.04, sigma_ln=0.9)
    scene = _scene_from_arrays(tmp_path, sc)
    for layout in ("wh3", "hw3"):
        ref = scene.render_image_hip(1, layout=layout).clone()
        n_lead = strips.tiles_along(w if layout == "wh3" else h, tile)
        n_other = strips.tiles_along(h if layout == "wh3" else w, tile)
        for parts in (2, 4, 16):
            bounds = strips.substrip_bounds(0, n_lead, parts)
            evs = []
            for _ in range(3):                  # cold frame, then frames with hints
                got = scene.render_image_hip(1, layout=layout, substrips=bounds, substrip_events=evs)
                assert len(evs) == parts
                torch.cuda.synchronize()
                assert torch.equal(got, ref), (layout, parts)
        # a strip window (what a rank renders), parts that do not divide it, one part empty
        t0, t1 = n_lead // 3, n_lead // 3 + 7
        window = (t0, t1, 0, n_other) if layout == "wh3" else (0, n_other, t0, t1)
        rows = (t1 - t0) * tile
        shape = (rows, h, 3) if layout == "wh3" else (rows, w, 3)
        origin = (t0 * tile, 0) if layout == "wh3" else (0, t0 * tile)
        whole = torch.empty(shape, dtype=torch.float32, device="cuda:0")
        scene.render_image_hip(1, layout=layout, tile_window=window, out=whole, out_origin=origin)
        assert torch.equal(whole, ref[t0 * tile:t1 * tile])
        bounds = [t0, t0 + 3, t0 + 3, t0 + 5, t1]
        out = torch.full(shape, -1.0, dtype=torch.float32, device="cuda:0")
        evs = []
        side = torch.cuda.Stream()
        seen = torch.empty_like(out)
        scene.render_image_hip(1, layout=layout, tile_window=window, out=out, out_origin=origin, substrips=bounds,
                               substrip_events=evs)
        for k in range(4):                      # another stream copies part k as soon as event k allows it
            a, b = (bounds[k] - t0) * tile, (bounds[k + 1] - t0) * tile
            evs[k].wait_on(side)
            with torch.cuda.stream(side):
                seen[a:b].copy_(out[a:b])
        torch.cuda.synchronize()
        assert torch.equal(out, whole) and torch.equal(seen, whole), layout
    # a rule set without a partial launch
    ref = scene.render_image_hip(1, semantics="std_3dgs").clone()
    evs = []
    got = scene.render_image_hip(1, semantics="std_3dgs", substrips=strips.substrip_bounds(0, strips.tiles_along(w, tile, "std_3dgs"), 3),
                                 substrip_events=evs)
    evs[2].synchronize()
    assert torch.equal(got, ref)
    # bounds that do not span the window are refused
    with pytest.raises(_ffi_error()):
        scene.render_image_hip(1, substrips=[0, 5, 9])


def _ffi_error():
    from intro_to_gaussian_splatting_amd import _ffi
    return _ffi.GsxError


# ----------------------------------------------------------------------------- error behaviour

def test_wrong_dtype_and_device_raise(tmp_path):
    _need_gpu()
    g = load_golden("small_64x48_n300")
    scene = _scene_from_golden(tmp_path, g)
    scene.gaussians.scales = scene.gaussians.scales.double()
    with pytest.raises(TypeError):
        scene.render_image(1)
    scene.gaussians.scales = scene.gaussians.scales.float().cpu()
    with pytest.raises(ValueError):
        scene.render_image(1)


def test_workspace_growth_retry(tmp_path):
    _need_gpu()
    from intro_to_gaussian_splatting_amd import gaussian_scene
    from intro_to_gaussian_splatting_amd.synthetic import make_scene

    sc = make_scene(500, 256, 256, seed=21)
    sc["scales"][:] *= 40.0                      # every splat covers most tiles: D >> 8 N
    scene = _scene_from_arrays(tmp_path, sc)
    gaussian_scene._WORKSPACE.buffers.clear()
    _, ref, inst = _oracle_frame(scene, sc)
    assert inst > 8 * 500 + 4096
    stats = {}
    img = scene.render_image_hip(1, stats=stats)
    assert stats["n_instances"] == inst
    assert np.max(np.abs(img.cpu().numpy() - ref)) <= PIXEL_TOL


def test_speculative_frames_equal_synchronised_frames(tmp_path):
    """GSX_FLAG_NO_SYNC: pair list sized by the previous frame's count, counts confirmed later."""
    _need_gpu()
    g = load_golden("c1_256x256_n2000")
    scene = _scene_from_golden(tmp_path, g)
    ref = scene.render_image_hip(1)                       # synchronising path; learns the count
    st = {}
    a = scene.render_image_hip(1, no_sync=True, stats=st)
    assert st.get("speculative") is True
    b = scene.render_image_hip(1, no_sync=True, layout="hw3")
    assert scene.confirm_frames() == 0
    assert torch.equal(a, ref) and torch.equal(b.permute(1, 0, 2), ref)
    # a workspace that is too small drops pairs on the device; confirm_frames notices and re-renders
    from intro_to_gaussian_splatting_amd import gaussian_scene
    from intro_to_gaussian_splatting_amd.synthetic import make_scene

    sc = make_scene(500, 256, 256, seed=21)
    sc["scales"][:] *= 40.0                      # every splat covers most tiles: D >> 8 N + 4096
    big = _scene_from_arrays(tmp_path / "big", sc)
    gaussian_scene._WORKSPACE.buffers.clear()
    c = big.render_image_hip(1, no_sync=True)    # first frame of the scene, enqueued blind
    assert big.confirm_frames() == 1
    assert torch.equal(c, big.render_image_hip(1)) and big._last_instances > 8 * 500 + 4096
    d = big.render_image_hip(1, no_sync=True)    # the workspace has grown: no miss any more
    assert big.confirm_frames() == 0 and torch.equal(c, d)
    # strips speculate too, each on the count of its own window
    strip = torch.empty((64, 256, 3), device="cuda:0")
    for _ in range(2):
        st = {}
        scene.render_image_hip(1, tile_window=(4, 8, 0, 15), out=strip, out_origin=(64, 0), no_sync=True, stats=st)
    assert st.get("speculative") is True and scene.confirm_frames() == 0
    assert torch.equal(strip, ref[64:128])
    assert scene._last_instances == 7379


def test_frames_in_flight_on_several_streams(tmp_path):
    """Frames of the same scene enqueued on different HIP streams (own scratch per stream) overlap
    on the GPU and must still be the single-stream frame, bit for bit."""
    _need_gpu()
    from intro_to_gaussian_splatting_amd.synthetic import make_scene

    sc = make_scene(200_000, 640, 480, seed=5)
    scene = _scene_from_arrays(tmp_path, sc)
    ref = scene.render_image_hip(1)
    streams = [torch.cuda.Stream("cuda:0") for _ in range(3)]
    outs = [torch.empty_like(ref) for _ in range(9)]
    torch.cuda.synchronize()
    for i, out in enumerate(outs):
        with torch.cuda.stream(streams[i % 3]):
            scene.render_image_hip(1, out=out, no_sync=(i % 2 == 0))
    torch.cuda.synchronize()
    assert scene.confirm_frames() == 0
    for out in outs:
        assert torch.equal(out, ref)


@pytest.mark.parametrize("name", ["small_64x48_n300", "cull_96x80_n400", "c1_256x256_n2000"])
def test_cuda_kernel_semantics_against_its_cpu_restatement(tmp_path, name):
    """render_image_cuda semantics (splat/c/render.cu).  Parity unpinned by the reference (its kernel
    cannot run here); checked against oracle/raster_cpu.c::orc_render_cuda_semantics.  The break at
    T(1-alpha) < 0.001 is a discontinuity of size <= 0.1, so a handful of pixels may legitimately
    differ when a test value lands within rounding of the threshold."""
    _need_gpu()
    from oracle import c_oracle

    g = load_golden(name)
    scene = _scene_from_golden(tmp_path, g)
    w, h = int(g["width"]), int(g["height"])
    img = scene.render_image_cuda(1)
    assert tuple(img.shape) == (h, w, 3)
    ref = c_oracle.render_cuda_semantics(golden_preprocessed(g), w, h)
    diff = np.abs(img.cpu().numpy() - ref).max(axis=2)
    n_over = int((diff > PIXEL_TOL).sum())
    print("ref_cuda %s: %d of %d pixels above %g (largest %.3g), median %.3g" % (
        name, n_over, diff.size, PIXEL_TOL, float(diff.max()), float(np.median(diff))))
    # observed on MI355X (round 2): 0 pixels above tolerance for all three fixtures; the bound leaves room for
    # a few 0.001-threshold flips, not for a regression
    assert n_over <= max(2, int(1e-4 * diff.size)) and np.median(diff) < 1e-6
    assert ref[h - 16:, :, :].any() or ref[:, w - 16:, :].any() or name.startswith("small")  # edge tiles are rendered
    # the whole-path entry with the same semantics gives the same frame, and other tile sizes too
    full = scene.render_image_hip(1, layout="hw3", semantics="ref_cuda")
    assert torch.equal(full, img)
    t8 = scene.render_image_hip(1, tile_size=8, layout="hw3", semantics="ref_cuda")
    assert torch.equal(t8, img)                                    # tile size is invisible in this mode
    cpu_sem = scene.render_image_hip(1, layout="hw3")
    assert not torch.equal(cpu_sem, img)                           # and the two semantics do differ
    # the reference's own call sequence (splat/gaussian_scene.py:263-285) through the compile_cuda_ext() shim
    pre = scene.preprocess(1)
    ext = scene.compile_cuda_ext()
    via_ext = ext.render_image(scene.images[1].height, scene.images[1].width, 16, pre.points.contiguous(),
                               pre.colors.contiguous(), pre.inverse_covariance_2d.contiguous(), pre.min_x.contiguous(),
                               pre.max_x.contiguous(), pre.min_y.contiguous(), pre.max_y.contiguous(),
                               pre.sigmoid_opacity.contiguous())
    torch.cuda.synchronize()
    assert tuple(via_ext.shape) == (h, w, 3) and torch.equal(via_ext, img)


def test_covariance_3d_method(tmp_path, golden):
    """Gaussians.get_3d_covariance_matrix (splat/gaussians.py:54-69) on the GPU: bit-identical to the
    reference's own matrices (and to the oracle, which has the same operation order)."""
    _need_gpu()
    from oracle import cpu_ref

    g = golden
    scene = _scene_from_golden(tmp_path, g)
    cov = scene.gaussians.get_3d_covariance_matrix().cpu().numpy()
    assert np.array_equal(cov, cpu_ref.covariance_3d(g["scales"], g["quaternions"]))
    ref = g["covariance_3d"]
    assert cov.shape == ref.shape and np.array_equal(cov.view(np.uint32), ref.view(np.uint32))


def test_points_projection_helper_against_the_references_output(tmp_path, golden):
    """GaussianScene.render_points_image (splat/gaussian_scene.py:44-51 -> splat/image.py:72-89): all three
    columns (x_pix, y_pix, ndc_z) and the colours against what the REFERENCE returned for the same scene."""
    _need_gpu()
    g = golden
    scene = _scene_from_golden(tmp_path, g)
    pts, cols = scene.render_points_image(1)
    ref = g["points_image_xyz"]
    assert tuple(pts.shape) == ref.shape and pts.shape[0] == int(g["in_view"].sum())
    got = pts.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))         # pixels and ndc z: the reference's bits
    assert np.array_equal(cols.cpu().numpy(), g["points_image_colors"])
    # the method of the camera object is the same call (splat/image.py:72-89)
    p2, c2 = scene.images[1].project_point_to_camera_perspective_projection(scene.gaussians.points, scene.gaussians.colors)
    assert torch.equal(p2, pts) and torch.equal(c2, cols)


def test_get_2d_covariance_wrapper_against_the_references_output(tmp_path, golden):
    """GaussianScene.get_2d_covariance(image_idx, points, covariance_3d) (splat/gaussian_scene.py:53-68) on the
    in-view points and the reference's own 3D covariances: the reference's result bit for bit, and bit-identical
    to the Sigma2D the stage-1 kernel computes inline."""
    _need_gpu()
    g = golden
    scene = _scene_from_golden(tmp_path, g)
    vis = torch.from_numpy(g["in_view"]).to("cuda:0")
    pts = scene.gaussians.points[vis]
    cov3 = torch.from_numpy(g["covariance_3d"]).to("cuda:0")[vis]
    out = scene.get_2d_covariance(1, pts, cov3)
    ref = g["get_2d_covariance"]
    assert tuple(out.shape) == ref.shape
    assert np.array_equal(out.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    # with the kernel's own 3D covariances it is what preprocess() stores (depth-sorted there)
    mine = scene.get_2d_covariance(1, pts, scene.gaussians.get_3d_covariance_matrix()[vis])
    pre = scene.preprocess(1)
    lookup = torch.full((g["points"].shape[0],), -1, dtype=torch.long, device="cuda:0")
    lookup[torch.nonzero(vis).squeeze(1)] = torch.arange(int(vis.sum()), device="cuda:0")
    assert torch.equal(mine[lookup[scene.last_order.long()]], pre.covariance_2d)


def test_a_stopped_pixel_stays_stopped_when_a_later_alpha_is_not_finite(tmp_path):
    """The reference RETURNS once T(1-alpha) < 1e-6 (gaussian_scene.py:166-167); the kernels instead keep a
    stopped pixel at T = 0.  A later Gaussian whose conic overflowed (inf / NaN inverse covariance, bounding
    box over the whole frame) must not revive it as 0 * inf = NaN.  `dense` fixture (a quarter of its pixels
    stop) + three such Gaussians appended at the far end, through the stage-2 entry point; the C restatement
    breaks out of the loop like the reference."""
    _need_gpu()
    from intro_to_gaussian_splatting_amd.gaussian_scene import render_preprocessed
    from oracle import c_oracle, cpu_ref

    g = load_golden("dense_48x48_n1500")
    pre = golden_preprocessed(g)
    w, h = int(g["width"]), int(g["height"])
    extra = 3
    inf, nan = np.float32(np.inf), np.float32(np.nan)
    cat = lambda a, b: np.concatenate([a, np.asarray(b, np.float32).reshape((extra,) + a.shape[1:])])   # noqa: E731
    # exponent = -1/2 d Q d^T.  First Q = NaN: every pixel that is still live turns NaN on both sides (the
    # reference's `NaN < 1e-6` is false, it accumulates).  Then Q = -inf (alpha = +inf) and a mixed one: a
    # stopped pixel computes 0 * inf = NaN in the kernel's arithmetic and must still keep its colour.
    bad_q = np.array([[[nan, nan], [nan, nan]], [[-inf, 0], [0, -inf]], [[-inf, inf], [-inf, -inf]]], np.float32)
    pre2 = cpu_ref.Preprocessed(
        cat(pre.points, [[24.3, 20.1]] * extra), cat(pre.colors, [[0.9, 0.8, 0.7]] * extra),
        cat(pre.covariance_2d, np.ones((extra, 2, 2))), cat(pre.depths, [99.0, 99.5, 99.9]),
        cat(pre.inverse_covariance_2d, bad_q), cat(pre.radius, [1e9] * extra), cat(pre.points_xy, [[24.3, 20.1]] * extra),
        cat(pre.min_x, [-inf] * extra), cat(pre.min_y, [-inf] * extra), cat(pre.max_x, [inf] * extra),
        cat(pre.max_y, [inf] * extra), cat(pre.sigmoid_opacity, [[0.7]] * extra),
        np.concatenate([pre.order, pre.order.max() + 1 + np.arange(extra)]))
    ref, _, inst = c_oracle.render(pre2, w, h, 16)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")   # noqa: E731
    for generic in (False, True):
        st = {}
        if generic:     # the any-tile-size kernel at tile 8 covers the same pixels except the last 8-px rows
            img = render_preprocessed(h, w, 8, t(pre2.points), t(pre2.colors), t(pre2.inverse_covariance_2d), t(pre2.min_x),
                                      t(pre2.max_x), t(pre2.min_y), t(pre2.max_y), t(pre2.sigmoid_opacity), stats=st)
            ref8, _, _ = c_oracle.render(pre2, w, h, 8)
            got, want = img.cpu().numpy(), ref8
        else:
            img = render_preprocessed(h, w, 16, t(pre2.points), t(pre2.colors), t(pre2.inverse_covariance_2d), t(pre2.min_x),
                                      t(pre2.max_x), t(pre2.min_y), t(pre2.max_y), t(pre2.sigmoid_opacity), stats=st)
            assert st["n_instances"] == inst
            got, want = img.cpu().numpy(), ref
        # pixels that had stopped keep their finite colour; live pixels turn NaN / inf on both sides alike
        assert np.array_equal(np.isfinite(got), np.isfinite(want))
        fin = np.isfinite(want)
        assert (fin & (np.abs(want) > 0)).any() and (~fin).any()      # stopped (finite, coloured) and live (NaN) pixels
        assert np.max(np.abs(got[fin] - want[fin])) <= PIXEL_TOL


def test_spherical_harmonics_kernel_and_trained_ply_pipeline(tmp_path):
    """Build extension (the reference has no SH; parity unpinned): the SH kernel against the float64
    numpy statement of the published convention, DC-only SH == the pinned RGB path, and a trained
    .ply (log scales, SH, logit opacity) rendered end to end."""
    _need_gpu()
    from intro_to_gaussian_splatting_amd import GaussianScene, Gaussians, ply
    from intro_to_gaussian_splatting_amd.synthetic import make_scene, write_colmap_text
    from oracle import cpu_ref

    sc = make_scene(5000, 256, 256, seed=17)
    write_colmap_text(str(tmp_path), sc)
    rgb = sc["colors_0_255"] / 256.0
    rs = np.random.RandomState(2)
    for deg in (0, 1, 2, 3):
        k = (deg + 1) ** 2
        sh = np.zeros((5000, k, 3), np.float32)
        sh[:, 0, :] = (rgb - 0.5) / 0.28209479177387814
        sh[:, 1:, :] = rs.normal(0, 0.2, size=(5000, k - 1, 3))
        path = str(tmp_path / ("trained_%d.ply" % deg))
        ply.save_trained(path, sc["points"], sh, sc["scales"], sc["quaternions"], sc["opacity"])
        g = Gaussians.from_ply(path, device="cuda:0")
        scene = GaussianScene(str(tmp_path), g)
        cols = scene._colors(1).cpu().numpy()
        ref = cpu_ref.sh_to_rgb(sc["points"], sh, deg, scene.images[1].camera_center.cpu().numpy())
        assert np.max(np.abs(cols - ref)) <= 2e-6
        img = scene.render_image(1)
        # the colour is evaluated INSIDE the projection kernel (GsxParams.sh): same code as the standalone kernel,
        # so feeding that kernel's colours through the RGB path gives the same frame bit for bit
        g_cols = Gaussians.from_arrays(g.points.cpu().numpy(), np.zeros_like(sc["colors_0_255"]), g.scales.cpu().numpy(),
                                       g.quaternions.cpu().numpy(), g.opacity.cpu().numpy(), device="cuda:0")
        g_cols.colors = scene._colors(1).clone()
        assert torch.equal(GaussianScene(str(tmp_path), g_cols).render_image_hip(1).cpu(), img)
        # a tile window (a rank's strip) evaluates the colours of the Gaussians that survive its window test only,
        # from rows staged by list instead of by block: same pixels
        for x0, x1, y0, y1 in ((3, 9, 0, 15), (0, 15, 6, 7), (14, 15, 14, 15)):
            part = scene.render_image_hip(1, tile_window=(x0, x1, y0, y1)).cpu()
            assert torch.equal(part[x0 * 16:x1 * 16, y0 * 16:y1 * 16], img[x0 * 16:x1 * 16, y0 * 16:y1 * 16]), (deg, x0)
        # the same frame through the pinned RGB path with those colours (colors = rgb/256 convention)
        g_rgb = Gaussians.from_arrays(sc["points"], ref * 256.0, np.exp(np.log(sc["scales"])), sc["quaternions"],
                                      sc["opacity"], device="cuda:0")
        img_rgb = GaussianScene(str(tmp_path), g_rgb).render_image(1)
        assert torch.max(torch.abs(img - img_rgb)).item() <= 1e-5
        if deg == 0:
            g_plain = Gaussians.from_arrays(sc["points"], sc["colors_0_255"], np.exp(np.log(sc["scales"])),
                                            sc["quaternions"], sc["opacity"], device="cuda:0")
            img_plain = GaussianScene(str(tmp_path), g_plain).render_image(1)
            assert torch.max(torch.abs(img - img_plain)).item() <= 1e-5


@pytest.mark.parametrize("n,bits,key16", [(1, 32, 0), (63, 32, 0), (8192, 32, 0), (8193, 32, 0), (100_003, 32, 0),
                                          (1_000_000, 32, 0), (70_001, 13, 1), (3_000_017, 16, 1), (500_000, 9, 1),
                                          (40_000, 24, 0), (131_072, 32, 0), (131_073, 32, 0), (131_073, 11, 1),
                                          (524_288, 32, 0), (524_289, 16, 1)])   # 64 / 65 count workgroups: last pass
                                                                                # without / first with a row-scan launch
def test_radix_sort_is_a_stable_sort(n, bits, key16):
    """gsx_sort.hip against torch's stable sort: same keys, same permutation (stability = the tie
    rule of the depth order and the reason the tile sort keeps depth order), incl. a device count."""
    _need_gpu()
    import ctypes

    from intro_to_gaussian_splatting_amd import _ffi

    lib = _ffi.load_test_hooks()
    fn = lib.gsx_debug_sort_pairs
    gen = torch.Generator(device="cuda:0").manual_seed(n)
    hi = (1 << bits) - 1
    # few distinct values in the low byte and many duplicates overall: exercises ties hard
    keys64 = torch.randint(0, min(hi, 5000) + 1, (n,), generator=gen, device="cuda:0", dtype=torch.int64)
    if bits > 16:
        keys64 = keys64 * 65537 % (hi + 1)
    vals = torch.arange(n, device="cuda:0", dtype=torch.int32)
    # two ping-pong arrays + the sort's digit table: gsx_workspace_bytes(n, .., n) holds more than that
    scratch = torch.empty(lib.gsx_workspace_bytes(n, 16, 16, 16, n), dtype=torch.uint8, device="cuda:0")
    for live in (n, max(1, (2 * n) // 3)):
        keys = keys64.to(torch.int16 if key16 else torch.int32).clone()
        v = vals.clone()
        count = torch.tensor([live], dtype=torch.int32, device="cuda:0")
        rc = fn(keys.data_ptr(), v.data_ptr(), n, bits, key16, count.data_ptr() if live != n else None,
                scratch.data_ptr(), scratch.numel(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0, lib.gsx_last_error()
        torch.cuda.synchronize()
        ref_k, ref_p = torch.sort(keys64[:live], stable=True)
        got_k = keys[:live].to(torch.int64) & (0xFFFF if key16 else 0xFFFFFFFF)
        assert torch.equal(got_k, ref_k)
        assert torch.equal(v[:live].to(torch.int64), ref_p)


@pytest.mark.parametrize("width,height,tile,n", [(250, 190, 16, 3000), (333, 257, 16, 2000), (600, 500, 2, 1500),
                                                 (130, 70, 7, 800)])
def test_awkward_frame_sizes_and_many_tiles(tmp_path, width, height, tile, n):
    """Frames that are not multiples of the tile (unaligned rows -> the scalar store path), and a
    frame with more than 65536 tiles (600x500 at tile 2: 74 451 tiles -> 32-bit tile ids)."""
    _need_gpu()
    from intro_to_gaussian_splatting_amd.strips import tiles_along
    from intro_to_gaussian_splatting_amd.synthetic import make_scene

    sc = make_scene(n, width, height, seed=width)
    scene = _scene_from_arrays(tmp_path, sc)
    _, ref, inst = _oracle_frame(scene, sc, tile)
    for layout in ("wh3", "hw3"):
        stats = {}
        img = scene.render_image_hip(1, tile_size=tile, layout=layout, stats=stats)
        assert stats["n_instances"] == inst and stats["n_tiles"] == tiles_along(width, tile) * tiles_along(height, tile)
        got = img.cpu().numpy() if layout == "wh3" else img.permute(1, 0, 2).cpu().numpy()
        assert got.shape == (width, height, 3)
        assert np.max(np.abs(got - ref)) <= PIXEL_TOL


def test_captured_frame_replays_in_a_hip_graph(tmp_path):
    """A no-sync frame has no host dependency, so the whole frame records into a hipGraph; a replay
    renders from the current Gaussian tensors."""
    _need_gpu()
    from intro_to_gaussian_splatting_amd.synthetic import make_scene

    sc = make_scene(50_000, 640, 480, seed=8)
    scene = _scene_from_arrays(tmp_path, sc)
    ref = scene.render_image_hip(1)
    frame = scene.capture_frame(1)
    frame.out.zero_()
    frame.replay()
    assert torch.equal(frame.confirm(), ref)
    # parameters changed in place (same tensors): the replay sees them
    scene.gaussians.colors.mul_(0.5)
    frame.replay()
    half = scene.render_image_hip(1)
    assert torch.equal(frame.confirm(), half) and not torch.equal(half, ref)
    assert scene.confirm_frames() == 0


def test_captured_frame_owns_its_capacity_scratch_and_count_slot(tmp_path):
    """capture_frame(headroom=h) records launches sized for h x the pair count of the captured view, in a
    scratch buffer and a pinned count slot that belong to that frame alone: later frames of the scene (more
    than the 256 slots of the shared ring) must not disturb them, and two captured frames replayed on
    different streams at the same time must not share scratch."""
    _need_gpu()
    import ctypes

    from intro_to_gaussian_splatting_amd import _ffi
    from intro_to_gaussian_splatting_amd.synthetic import make_scene

    sc = make_scene(30_000, 480, 320, seed=12)
    scene = _scene_from_arrays(tmp_path, sc)
    st = {}
    ref = scene.render_image_hip(1, stats=st).clone()
    d = st["n_instances"]
    a = scene.capture_frame(1, headroom=2.0)
    b = scene.capture_frame(1, headroom=1.0)
    assert a.capacity >= 2 * d and d <= b.capacity < a.capacity
    assert a._workspace.data_ptr() != b._workspace.data_ptr() and a._pinned.data_ptr() != b._pinned.data_ptr()
    a.replay()
    a.confirm()
    got = ctypes.cast(ctypes.c_void_p(a._pinned.data_ptr()), ctypes.POINTER(_ffi.GsxFrameStats)).contents
    assert got.reserved >= 2 * d and got.n_instances == d
    out = torch.empty_like(ref)
    for _ in range(300):                      # wraps the scene's 256-slot ring of pinned count slots
        scene.render_image_hip(1, out=out, no_sync=True)
    assert scene.confirm_frames() == 0
    got = ctypes.cast(ctypes.c_void_p(a._pinned.data_ptr()), ctypes.POINTER(_ffi.GsxFrameStats)).contents
    assert got.reserved >= 2 * d and got.n_instances == d          # untouched by the ring
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(5):
        with torch.cuda.stream(s1):
            a.replay()
        with torch.cuda.stream(s2):
            b.replay()
    torch.cuda.synchronize()
    assert torch.equal(a.confirm(), ref) and torch.equal(b.confirm(), ref)


def test_c_abi_rejects_bad_arguments(tmp_path):
    _need_gpu()
    from intro_to_gaussian_splatting_amd import _ffi

    g = load_golden("small_64x48_n300")
    scene = _scene_from_golden(tmp_path, g)
    small = torch.empty((16, 48, 3), device="cuda:0")
    with pytest.raises(_ffi.GsxError, match="does not fit"):       # window larger than the strip buffer
        scene.render_image_hip(1, tile_window=(0, 3, 0, 2), out=small, out_origin=(0, 0))
    with pytest.raises(_ffi.GsxError, match="rejected the sizes"):
        scene.render_image_hip(1, tile_size=0)
    with pytest.raises(KeyError):
        scene.render_image_hip(1, semantics="no_such_rules")
    lib = _ffi.load()
    p = _ffi.default_params()
    p.layout = 7
    out = torch.empty((64, 48, 3), device="cuda:0")
    ws = torch.empty(1 << 20, dtype=torch.uint8, device="cuda:0")
    cam = scene.images[1].gsx_camera()
    import ctypes
    rc = lib.gsx_render_forward(ctypes.byref(cam), None, None, None, None, None, 0, 16, out.data_ptr(), ctypes.byref(p),
                                None, ws.data_ptr(), ws.numel(), None)
    assert rc == _ffi.GSX_ERR_INVALID_ARGUMENT and b"layout" in lib.gsx_last_error()
    rc = lib.gsx_render_forward(ctypes.byref(cam), None, None, None, None, None, 5, 16, out.data_ptr(), None, None,
                                ws.data_ptr(), ws.numel(), None)
    assert rc == _ffi.GSX_ERR_INVALID_ARGUMENT and b"NULL" in lib.gsx_last_error()
    tiny = torch.empty(256, dtype=torch.uint8, device="cuda:0")
    rc = lib.gsx_render_forward(ctypes.byref(cam), out.data_ptr(), out.data_ptr(), out.data_ptr(), out.data_ptr(),
                                out.data_ptr(), 100, 16, out.data_ptr(), None, None, tiny.data_ptr(), tiny.numel(), None)
    assert rc == _ffi.GSX_ERR_WORKSPACE_TOO_SMALL


def test_scene_built_like_the_notebooks(tmp_path):
    """cpu_render.ipynb builds Gaussians(points, rgb) with the constructor defaults (scale 0.001,
    identity rotation, opacity logit(0.9999)): tiny splats, so the eigenvalue floor (0.1) and the
    determinant floor (1e-3) set every radius and conic.  52 363 points like Treehill's sparse cloud."""
    _need_gpu()
    from intro_to_gaussian_splatting_amd import GaussianScene, Gaussians
    from intro_to_gaussian_splatting_amd.synthetic import make_scene, write_colmap_text
    from oracle import c_oracle

    sc = make_scene(52_363, 256, 256, seed=100)
    write_colmap_text(str(tmp_path), sc)
    g = Gaussians(torch.from_numpy(sc["points"]), torch.from_numpy(np.floor(sc["colors_0_255"])), device="cuda:0")
    scene = GaussianScene(str(tmp_path), g)
    cam = _oracle_cam(scene)
    pre = c_oracle.preprocess(sc["points"], g.colors.cpu().numpy(), g.scales.cpu().numpy(), g.quaternions.cpu().numpy(),
                              g.opacity.cpu().numpy(), cam)
    ref, _, inst = c_oracle.render(pre, 256, 256, 16)
    stats = {}
    img = scene.render_image_hip(1, stats=stats)
    assert stats["n_instances"] == inst and stats["n_visible"] == 52_363
    assert np.max(np.abs(img.cpu().numpy() - ref)) <= PIXEL_TOL
    gp = scene.preprocess(1)
    assert torch.all(gp.radius == 2.0)                          # ceil(3 sqrt(mid + sqrt(0.1))) with a ~1e-6 px covariance
    assert ref.max() > 0.7 and ref.max() < 0.7311 * 255 / 256 + 0.3   # double sigmoid: sigma(sigma(9.21)) = 0.7311 per splat


@pytest.mark.parametrize("seed", range(12))
def test_randomised_small_scenes(tmp_path, seed):
    """Seeded sweep over frame sizes, tile sizes, layouts, semantics, tile windows, culled and
    degenerate splats; every frame against the C restatement."""
    _need_gpu()
    from intro_to_gaussian_splatting_amd.strips import tiles_along
    from intro_to_gaussian_splatting_amd.synthetic import make_scene
    from oracle import c_oracle

    rs = np.random.RandomState(1000 + seed)
    width, height = int(rs.randint(20, 400)), int(rs.randint(20, 300))
    tile = int(rs.choice([1, 2, 3, 8, 16, 16, 16, 17, 32, 64]))
    n = int(rs.choice([1, 2, 17, 100, 1000, 5000]))
    sc = make_scene(n, width, height, seed=seed, behind_fraction=float(rs.choice([0.0, 0.3])))
    sc["scales"] *= np.float32(rs.choice([0.05, 1.0, 1.0, 6.0]))
    if n >= 17:
        sc["scales"][3] = 0.0                       # degenerate: zero covariance -> the floors decide
        sc["quaternions"][5] = 0.0                  # zero quaternion -> NaN rotation, like the reference
        sc["opacity"][7] = 80.0
        sc["opacity"][9] = -80.0
    layout = str(rs.choice(["wh3", "hw3"]))
    scene = _scene_from_arrays(tmp_path, sc)
    cam = _oracle_cam(scene)
    pre = c_oracle.preprocess(sc["points"], scene.gaussians.colors.cpu().numpy(), sc["scales"], sc["quaternions"],
                              sc["opacity"], cam)
    ntx, nty = tiles_along(width, tile), tiles_along(height, tile)
    window = None
    if ntx > 1 and nty > 1 and rs.rand() < 0.5:
        a, b = sorted(rs.randint(0, ntx + 1, 2))
        c, d = sorted(rs.randint(0, nty + 1, 2))
        if a < b and c < d:
            window = (int(a), int(b), int(c), int(d))
    ref, _, inst = c_oracle.render(pre, width, height, tile, window=window)
    stats = {}
    img = scene.render_image_hip(1, tile_size=tile, layout=layout, tile_window=window, stats=stats)
    got = img.cpu().numpy() if layout == "wh3" else img.permute(1, 0, 2).cpu().numpy()
    finite = np.isfinite(ref)
    assert np.array_equal(np.isfinite(got), finite)
    assert np.max(np.abs(got[finite] - ref[finite]), initial=0.0) <= PIXEL_TOL
    assert stats["n_visible"] == pre.points.shape[0]
    if window is None:
        assert stats["n_instances"] == inst
    # the CUDA-kernel semantics on the same scene (whole frame) against its restatement
    img2 = scene.render_image_hip(1, tile_size=tile, layout="hw3", semantics="ref_cuda").cpu().numpy()
    ref2 = c_oracle.render_cuda_semantics(pre, width, height)
    ok = np.isfinite(ref2) & np.isfinite(img2)
    diff = np.abs(img2 - ref2)[ok]
    assert (diff > PIXEL_TOL).mean() < 2e-3 if diff.size else True


def test_camera_constants_equal_the_references_on_this_host(tmp_path, golden):
    """The camera constants are computed on the host; they must come out bit-equal to the ones the
    reference computed where the golden vectors were captured, also on the GPU box's CPU (whose
    torch.sqrt is not correctly rounded -- image.py takes the square root in double)."""
    _need_gpu()
    g = golden
    cam = _scene_from_golden(tmp_path, g).images[1].gsx_camera()
    assert np.array_equal(np.array(list(cam.world2view), np.float32).reshape(4, 4), g["world2view"])
    assert np.array_equal(np.array(list(cam.full_proj), np.float32).reshape(4, 4), g["full_proj_transform"])
    assert np.float32(cam.tan_fovx) == g["tan_fovX"][0] and np.float32(cam.tan_fovy) == g["tan_fovY"][0]
    assert np.float32(cam.fx) == g["f_x"][0] and np.float32(cam.fy) == g["f_y"][0]


def test_pair_count_beyond_32_bits_is_reported_not_wrapped(tmp_path):
    """17 000 frame-filling Gaussians x 262 144 tiles of 2x2 pixels = 4.46e9 pairs: more than the
    32-bit offsets hold.  The frame cannot be rendered, but the count that comes back must be the
    true one (a wrapped count could pass for a small frame)."""
    _need_gpu()
    from intro_to_gaussian_splatting_amd import _ffi
    from intro_to_gaussian_splatting_amd.synthetic import make_scene

    w = h = 1026
    sc = make_scene(17000, w, h, seed=2)
    sc["scales"] = np.full_like(sc["scales"], 50.0)            # every bounding box covers the frame
    scene = _scene_from_arrays(tmp_path, sc)
    lib = _ffi.load()
    dev, n, tensors = scene._inputs(1)
    cam = scene.images[1].gsx_camera()
    cap = 1 << 20
    nbytes = lib.gsx_workspace_bytes(n, w, h, 2, cap)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    out = torch.empty((w, h, 3), dtype=torch.float32, device=dev)
    st = _ffi.GsxFrameStats()
    import ctypes

    rc = lib.gsx_render_forward(ctypes.byref(cam), *[ctypes.c_void_p(t.data_ptr()) for t in tensors], n, 2,
                                ctypes.c_void_p(out.data_ptr()), None, ctypes.byref(st),
                                ctypes.c_void_p(ws.data_ptr()), nbytes,
                                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == _ffi.GSX_ERR_WORKSPACE_TOO_SMALL
    assert st.n_visible == 17000 and st.n_instances == 17000 * 512 * 512
    with pytest.raises(_ffi.GsxError):                          # the Python surface gives up cleanly
        scene.render_image_hip(1, tile_size=2)


@pytest.mark.parametrize("layout", ["wh3", "hw3"])
def test_strip_pipeline_frames_in_flight_on_one_gpu(tmp_path, layout):
    """strips.StripPipeline with 3 frames in flight on side streams (world size 1: the collective is a
    copy, the stream / event choreography is the real one): every submitted frame equals the plain
    render."""
    _need_gpu()
    from intro_to_gaussian_splatting_amd import strips
    from intro_to_gaussian_splatting_amd.synthetic import make_scene

    w, h = 320, 200
    sc = make_scene(30000, w, h, seed=21)
    scene = _scene_from_arrays(tmp_path, sc)
    ref = scene.render_image_hip(1, layout=layout)

    def render(window, out, origin):
        scene.render_image_hip(1, layout=layout, tile_window=window, out=out, out_origin=origin, no_sync=True)

    pipe = strips.StripPipeline(render, w, h, 16, layout, torch.device("cuda:0"), depth=3)
    for i in range(8):
        frame = pipe.submit()
        if i in (0, 3, 7):
            torch.cuda.synchronize()
            assert torch.equal(frame, ref), i
    torch.cuda.synchronize()
    assert scene.confirm_frames() == 0
    assert torch.equal(frame, ref)


@pytest.mark.parametrize("n,deg,skip", [(1, 3, 0), (255, 1, 0), (256, 2, 0), (257, 3, 0), (1000, 2, 1), (70001, 3, 0),
                                        (513, 0, 0)])
def test_sh_kernel_block_edges_and_unaligned_input(n, deg, skip):
    """gsx_sh_to_rgb stages 256 Gaussians' coefficients through LDS with 16-byte loads: block tails,
    every degree's row length (3, 12, 27, 48 floats) and a coefficient pointer that is not 16-byte
    aligned (a view that skips one Gaussian at degree 2)."""
    _need_gpu()
    import ctypes

    from intro_to_gaussian_splatting_amd import _ffi
    from oracle import cpu_ref

    lib = _ffi.load()
    rs = np.random.RandomState(n + deg)
    k = (deg + 1) ** 2
    pts = rs.normal(size=(n + skip, 3)).astype(np.float32)
    sh = rs.normal(0, 0.3, size=(n + skip, k, 3)).astype(np.float32)
    center = np.array([0.3, -0.2, 4.0], np.float32)
    d_pts = torch.from_numpy(pts).cuda()[skip:].contiguous()
    d_sh_all = torch.from_numpy(sh).cuda()
    d_sh = d_sh_all[skip:]                                   # a view: base pointer + skip * 12 k bytes
    assert d_sh.is_contiguous() and (skip == 0 or d_sh.data_ptr() % 16 != 0)
    out = torch.empty((n, 3), dtype=torch.float32, device="cuda:0")
    c = (ctypes.c_float * 3)(*center.tolist())
    rc = lib.gsx_sh_to_rgb(ctypes.c_void_p(d_pts.data_ptr()), ctypes.c_void_p(d_sh.data_ptr()), deg, n, c,
                           ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _ffi.check(rc)
    ref = cpu_ref.sh_to_rgb(pts[skip:], sh[skip:], deg, center)
    assert np.max(np.abs(out.cpu().numpy() - ref)) <= 3e-6


def test_captured_frame_follows_a_moving_camera(tmp_path):
    """capture_frame(movable_camera=True): the recorded projection kernel reads the camera from a device
    buffer (GsxParams.camera_device); set_camera() + replay() renders another view with the same graph."""
    _need_gpu()
    from intro_to_gaussian_splatting_amd.image import GaussianImage
    from intro_to_gaussian_splatting_amd.synthetic import make_scene

    sc = make_scene(20000, 320, 240, seed=31)
    scene = _scene_from_arrays(tmp_path, sc)
    first = scene.images[1]
    # two more poses of the same camera model
    from intro_to_gaussian_splatting_amd.colmap import read_camera_file, read_image_file
    cam_model = read_camera_file(str(tmp_path))[1]
    base = read_image_file(str(tmp_path))[1]
    for idx, (dq, dt) in {2: ((0.02, -0.03, 0.01), (0.3, -0.1, 0.2)), 3: ((-0.05, 0.02, 0.04), (-0.4, 0.2, -0.3))}.items():
        q = np.array(base.qvec, np.float64) + np.array((0.0,) + dq)
        t = np.array(base.tvec, np.float64) + np.array(dt)
        scene.images[idx] = GaussianImage(camera=cam_model, image=base._replace(qvec=q / np.linalg.norm(q), tvec=t),
                                          device="cuda:0")
    refs = {i: scene.render_image_hip(i).clone() for i in (1, 2, 3)}
    assert not torch.equal(refs[1], refs[2]) and not torch.equal(refs[2], refs[3])
    frame = scene.capture_frame(1, movable_camera=True, headroom=1.5)
    for i in (1, 2, 3, 1, 3):
        frame.set_camera(i)
        frame.replay()
        assert torch.equal(frame.confirm(), refs[i]), i
    baked = scene.capture_frame(1)
    with pytest.raises(RuntimeError, match="movable_camera"):
        baked.set_camera(2)
    assert first is scene.images[1]
    # a trained-.ply style scene (degree-3 spherical harmonics): the colour depends on the camera centre, which the
    # recorded projection kernel reads from the same device buffer
    from intro_to_gaussian_splatting_amd import GaussianScene, Gaussians, ply

    rs = np.random.RandomState(3)
    sh = np.zeros((20000, 16, 3), np.float32)
    sh[:, 0, :] = (sc["colors_0_255"] / 256.0 - 0.5) / 0.28209479177387814
    sh[:, 1:, :] = rs.normal(0, 0.25, size=(20000, 15, 3))
    path = str(tmp_path / "trained.ply")
    ply.save_trained(path, sc["points"], sh, sc["scales"], sc["quaternions"], sc["opacity"])
    sh_scene = GaussianScene(str(tmp_path), Gaussians.from_ply(path, device="cuda:0"))
    for idx in (2, 3):
        sh_scene.images[idx] = scene.images[idx]
    sh_refs = {i: sh_scene.render_image_hip(i).clone() for i in (1, 2, 3)}
    assert not torch.equal(sh_refs[1], refs[1])                      # the view-dependent part is there
    sh_frame = sh_scene.capture_frame(1, movable_camera=True, headroom=1.5)
    for i in (2, 1, 3, 3, 1):
        sh_frame.set_camera(i)
        sh_frame.replay()
        assert torch.equal(sh_frame.confirm(), sh_refs[i]), i


def test_hinted_frames_equal_frames_rendered_from_scratch(tmp_path):
    """GsxParams.hints (round 3): a frame takes the depth-sort splitters and the tile hand-out order from what the
    PREVIOUS frame of the view left in a small device buffer instead of computing them on its own critical path.
    Both are correct whatever their values -- equal depth keys share a bucket and every step is stable, the schedule
    is a permutation of the tiles -- so the hinted frame must equal the frame rendered from scratch BIT FOR BIT and
    report the same counts: for fresh hints, for stale ones (the camera changed between the frames; the scene was
    replaced by another), for a buffer that was never filled (zeroed: the stand-in splitters, index order), on a
    tile window, and inside a captured frame replayed while the camera moves."""
    _need_gpu()
    from intro_to_gaussian_splatting_amd import GaussianScene, Gaussians
    from intro_to_gaussian_splatting_amd.synthetic import make_scene, write_colmap_text

    w, h, n = 1280, 720, 400_000
    sc = make_scene(n, w, h, seed=21)
    write_colmap_text(str(tmp_path), sc)
    # a second camera for the same scene: slightly rotated and moved back
    with open(tmp_path / "images.txt", "a") as fid:
        q = np.array([0.95, -0.25, 0.15, 0.05]); q = q / np.linalg.norm(q)
        fid.write("2 %r %r %r %r 0.1 0.8 4.2 1 other.jpg\n1.0 2.0 -1\n" % tuple(float(v) for v in q))
    g = Gaussians.from_arrays(sc["points"], sc["colors_0_255"], sc["scales"], sc["quaternions"], sc["opacity"], device="cuda:0")
    scene = GaussianScene(str(tmp_path), g)
    ref = {}
    for cam in (1, 2):
        st = {}
        ref[cam] = (scene.render_image_hip(cam, use_hints=False, stats=st).clone(), st["n_instances"], st["n_visible"])
        assert st["n_instances"] > 5 * 3520                     # (far more than two tiles per SIMD's worth of lists)
    for rep in range(3):                                         # frame 0 fills the buffer, frames 1, 2 use it
        st = {}
        img = scene.render_image_hip(1, stats=st)
        assert torch.equal(img, ref[1][0]) and (st["n_instances"], st["n_visible"]) == ref[1][1:], rep
    # stale hints: the buffer of camera 1's view, used for camera 2's frame (same frame size) -- and back
    key1 = [k for k in scene._hints if k[0][0] == 1][0]
    slot = scene._hints[key1]
    assert slot[1] is True
    hdr = slot[0][:16].view(torch.int32).cpu().numpy()
    assert hdr[0] == 256 and hdr[2] == hdr[3] > 2048, hdr       # splitters, list lengths and a schedule are on file
    scene._hints[(key1[0][:0] + (2,) + key1[0][1:], key1[1])] = slot
    for rep in range(2):
        img2 = scene.render_image_hip(2)
        assert torch.equal(img2, ref[2][0]), rep
    assert torch.equal(scene.render_image_hip(1), ref[1][0])     # camera 1 again, with what camera 2's frames left
    # a buffer nobody filled, declared valid: zeros -> stand-in splitters, no schedule; still the same frame
    slot[0].zero_()
    assert torch.equal(scene.render_image_hip(1), ref[1][0])
    # another scene through the same buffer (fewer, larger splats): its hints are stale in every respect
    sc2 = make_scene(150_000, w, h, seed=22, sigma_scale=2.0)
    g2 = Gaussians.from_arrays(sc2["points"], sc2["colors_0_255"], sc2["scales"], sc2["quaternions"], sc2["opacity"], device="cuda:0")
    scene2 = GaussianScene(str(tmp_path), g2)
    fresh2 = scene2.render_image_hip(1, use_hints=False).clone()
    scene2._hints[key1] = slot
    assert torch.equal(scene2.render_image_hip(1), fresh2) and torch.equal(scene2.render_image_hip(1), fresh2)
    # a tile window keeps hints of its own
    win = (10, 50, 3, 40)
    a = scene.render_image_hip(1, tile_window=win, use_hints=False).clone()
    for rep in range(3):
        assert torch.equal(scene.render_image_hip(1, tile_window=win), a)
    # a captured frame: its graph reads and refreshes its own hints on every replay, also while the camera moves
    frame = scene.capture_frame(1, movable_camera=True)
    for cam in (1, 2, 2, 1, 1):
        frame.set_camera(cam)
        frame.out.fill_(5.0)
        frame.replay()
        assert torch.equal(frame.confirm(), ref[cam][0]), cam


def test_multi_gpu_path_meets_rccl_with_a_world_of_one():
    """No multi-GPU node has been available to any round: every collective of the strip path had only ever run on gloo.
    ``bench.py --gpus 1 --dist-preflight`` runs THAT code -- dist.init_process_group("nccl") (= RCCL), the overlapped
    sub-strip gather (strips.render_overlapped), the plain gather, StripPipeline, the barriers / reductions / broadcast of
    the timing protocol, the `distributed` block -- with a process group of one rank on this GPU: communicator creation,
    stream / event ordering and every call have then executed against the real backend at least once."""
    _need_gpu()
    import json
    import subprocess
    import sys

    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dist-preflight", "--workload", "c2",
                          "--steps", "3", "--warmup", "1", "--repeats", "2", "--settle-ms", "10"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    line = json.loads([ln for ln in run.stdout.splitlines() if ln.startswith("{")][-1])
    d = line["distributed"]
    print(json.dumps(d))
    assert d["backend"] == "nccl" and d["world_size"] == 1 and line["n_gpus"] == 1
    assert line["strips_equal_single_gpu"] is True
    assert d["gather"].startswith("overlapped"), d            # the overlapped path passed its self-check against the lone frame
    assert d["preflight"]["gather_one_rank"] is True, d["preflight"]
    assert d["preflight"]["send_recv_to_self"] is True, d["preflight"]
    assert d["frames_in_flight_path"].startswith("strips.StripPipeline")


@pytest.mark.parametrize("name", ["ties_64x64_n400", "trainedlike_128x128_n3000", "cull_96x80_n400", "fewvisible_48x48_n9"])
def test_spatially_ordered_scene_renders_the_same_frame_bit_for_bit(tmp_path, name):
    """Gaussians.spatially_ordered() (GsxParams.original_index): the rows of the parameter arrays reordered along a Morton
    curve, everything filed under the original index -- same frame bit for bit (whole frame, a tile window, a captured
    frame), same permutation (``last_order`` holds ORIGINAL indices; on the tie fixture 367 of 400 Gaussians share a depth:
    they still composite in original-index order), same stage-1 arrays, same counts."""
    _need_gpu()
    from intro_to_gaussian_splatting_amd import GaussianScene

    g = load_golden(name)
    scene = _scene_from_golden(tmp_path, g)
    ordered = GaussianScene(str(tmp_path), scene.gaussians.spatially_ordered())
    oi = ordered.gaussians.original_index.cpu().numpy()
    if oi.size > 3:
        assert not np.array_equal(oi, np.arange(oi.size))
    tile = int(g["tile"])
    sa, sb = {}, {}
    a = scene.render_image_hip(1, tile_size=tile, stats=sa)
    b = ordered.render_image_hip(1, tile_size=tile, stats=sb)
    assert torch.equal(a, b)
    assert {k: sa[k] for k in ("n_visible", "n_instances", "n_kept", "n_redo")} == {k: sb[k] for k in ("n_visible", "n_instances", "n_kept", "n_redo")}
    assert np.max(np.abs(b.cpu().numpy() - g["image"])) <= (0.25 if name.startswith("ties") else PIXEL_TOL)
    pa, pb = scene.preprocess(1), ordered.preprocess(1)
    assert torch.equal(scene.last_order, ordered.last_order)
    for f in pa._fields:
        assert torch.equal(getattr(pa, f), getattr(pb, f)), f
    ntx = (int(g["width"]) - 1) // tile
    if ntx >= 2:
        win = (1, ntx, 0, -1)
        assert torch.equal(scene.render_image_hip(1, tile_size=tile, tile_window=win), ordered.render_image_hip(1, tile_size=tile, tile_window=win))
    frame = ordered.capture_frame(1, tile_size=tile)
    assert torch.equal(frame.replay(), a)
    frame.confirm()


def test_spatially_ordered_strips_of_a_100k_scene(tmp_path):
    """C2 (100k Gaussians, 1080p) in eight column strips from spatially ordered rows: every strip equals the unordered
    scene's strip, the strips tile the frame, and the SH path (degree 3, evaluated inside the windowed projection kernel
    from the survivors' coefficient rows) follows the permutation too."""
    _need_gpu()
    from intro_to_gaussian_splatting_amd import GaussianScene, strips
    from intro_to_gaussian_splatting_amd.synthetic import make_trained_like_scene

    sc = make_trained_like_scene(100_000, 1920, 1080, seed=3)
    scene = _scene_from_arrays(tmp_path, sc)
    for with_sh in (False, True):
        if with_sh:
            scene.gaussians.sh = torch.from_numpy(sc["sh"]).to("cuda:0").contiguous()
            scene.gaussians.sh_degree = int(sc["sh_degree"])
        ordered = GaussianScene(str(tmp_path), scene.gaussians.spatially_ordered())
        whole = scene.render_image_hip(1).clone()
        assert torch.equal(ordered.render_image_hip(1), whole)
        ntx, nty = strips.tiles_along(1920, 16), strips.tiles_along(1080, 16)
        for t0, t1 in strips.strip_plan(ntx, 8)[1]:
            win = (t0, t1, 0, nty)
            out = torch.empty(((t1 - t0) * 16, 1080, 3), dtype=torch.float32, device="cuda:0")
            ordered.render_image_hip(1, tile_window=win, out=out, out_origin=(t0 * 16, 0))
            assert torch.equal(out, whole[t0 * 16:t1 * 16]), (with_sh, t0, t1)
    # the Gaussians MOVE (in place, as an optimiser step or an edit would): the boxes the strip's projection drops blocks by
    # follow (Gaussians.current_block_bounds) -- a third of the scene shifted sideways by a quarter of the frame's width
    og = ordered.gaussians
    moved = torch.arange(0, 100_000, 3, device="cuda:0")
    shift = torch.tensor([1.5, 0.0, 0.0], device="cuda:0")
    scene.gaussians.points[moved] += shift
    og.points[og.row_of_index[moved].long()] += shift
    whole = scene.render_image_hip(1).clone()
    for t0, t1 in strips.strip_plan(ntx, 8)[1]:
        out = torch.empty(((t1 - t0) * 16, 1080, 3), dtype=torch.float32, device="cuda:0")
        ordered.render_image_hip(1, tile_window=(t0, t1, 0, nty), out=out, out_origin=(t0 * 16, 0))
        assert torch.equal(out, whole[t0 * 16:t1 * 16]), ("moved", t0, t1)


def test_a_frame_leaves_the_exact_depth_quantiles_as_the_next_frames_splitters(tmp_path):
    """The 256-bucket depth sort knows every kept key's rank: its bucket kernel leaves the NEXT frame's splitters in the hints
    buffer itself -- splitters[j] = the key of rank floor(j M / 256), the exact quantiles (round 6; before, 2 048 sampled keys
    were ranked by spare workgroups of the compositing launch: buckets of 0.5 .. 2x the mean, and the largest bucket is the
    bucket kernel's duration).  Read back here: 256 words, ascending, equal to the quantiles of the sorted depth keys -- and
    the frame that uses them is the same bit for bit."""
    _need_gpu()
    from intro_to_gaussian_splatting_amd.synthetic import make_scene

    sc = make_scene(60_000, 800, 608, seed=9)
    scene = _scene_from_arrays(tmp_path, sc)
    st = {}
    first = scene.render_image_hip(1, stats=st).clone()
    torch.cuda.synchronize()
    slots = [v for v in scene._hints._d.values()] if hasattr(scene._hints, "_d") else list(scene._hints.values())
    assert len(slots) == 1
    words = slots[0][0].view(torch.int32).cpu().numpy().view(np.uint32)
    assert words[0] == 256                                         # header[kHintSplitters]
    spl = words[64:64 + 256].astype(np.int64)
    assert spl[0] == 0 and np.all(np.diff(spl[1:]) >= 0)
    pre = scene.preprocess(1)
    depths = np.sort(pre.depths.cpu().numpy().view(np.uint32).astype(np.int64))
    m = int(st["n_kept"])
    if m == depths.size:                                            # (every visible Gaussian reaches a tile of this frame)
        want = depths[(np.arange(1, 256) * m) // 256]
        assert np.array_equal(spl[1:], want)
    assert torch.equal(scene.render_image_hip(1), first)
