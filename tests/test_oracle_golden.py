"""Pins the oracle (oracle/cpu_ref.py, oracle/raster_cpu.c) to golden vectors produced by the
reference itself (oracle/capture_golden.py -> tests/golden/*.npz).  CPU only."""
import numpy as np
import pytest

from conftest import golden_preprocessed, load_golden, oracle_camera
from oracle import c_oracle, cpu_ref

# Continuous fields may differ from the reference by float32 re-association inside its BLAS
# matmuls; integer-valued fields (radius, bbox) must match exactly on fixtures.
EXACT = ["radius", "min_x", "min_y", "max_x", "max_y", "colors"]
# field -> (rtol, atol): |a - b| <= atol + rtol |b|.  Pixel positions come out of (ndc + 1), a
# cancellation, so they get an absolute bound (1e-4 px on frames <= 256 px wide).
CLOSE = {"points": (0.0, 1e-4), "covariance_2d": (2e-4, 1e-6), "depths": (0.0, 2e-6),
         "inverse_covariance_2d": (2e-4, 1e-6), "sigmoid_opacity": (0.0, 2.4e-7)}


def _check_stage1(pre, g):
    assert np.array_equal(pre.order, g["order"]), "depth permutation differs from the reference's argsort"
    for f in EXACT:
        assert np.array_equal(getattr(pre, f), g["pre_" + f]), f
    for f, (rtol, atol) in CLOSE.items():
        a, b = getattr(pre, f), g["pre_" + f]
        assert np.all(np.abs(a - b) <= atol + rtol * np.abs(b)), f


def test_camera_constants_match_reference(golden):
    g = golden
    cam = cpu_ref.build_camera(g["qvec"], g["tvec"], g["fx"], g["fy"], int(g["width"]), int(g["height"]))
    assert np.array_equal(cam.world2view, g["world2view"])
    assert np.array_equal(cam.full_proj, g["full_proj_transform"])
    assert abs(cam.tan_fovx - g["tan_fovX"][0]) <= 1.2e-7
    assert abs(cam.tan_fovy - g["tan_fovY"][0]) <= 1.2e-7


def test_numpy_stage1_matches_reference(golden):
    g = golden
    pre = cpu_ref.preprocess(g["points"], g["colors"], g["scales"], g["quaternions"], g["opacity"], oracle_camera(g))
    assert pre.points.shape[0] == int(g["in_view"].sum())
    _check_stage1(pre, g)


def test_c_stage1_matches_reference_and_numpy_bitwise(golden):
    g = golden
    cam = oracle_camera(g)
    a = cpu_ref.preprocess(g["points"], g["colors"], g["scales"], g["quaternions"], g["opacity"], cam)
    b = c_oracle.preprocess(g["points"], g["colors"], g["scales"], g["quaternions"], g["opacity"], cam)
    _check_stage1(b, g)
    for f in a._fields:
        if f == "sigmoid_opacity":  # libm expf vs numpy exp: 1 ulp
            assert np.max(np.abs(getattr(a, f) - getattr(b, f))) <= 1.2e-7
        else:
            assert np.array_equal(getattr(a, f), getattr(b, f)), f


@pytest.mark.parametrize("name", ["small_64x48_n300", "small_80x64_n120_tile8", "cull_96x80_n400"])
def test_numpy_blend_given_reference_stage1(name):
    g = load_golden(name)
    img = cpu_ref.render_image(golden_preprocessed(g), int(g["width"]), int(g["height"]), int(g["tile"]))
    assert img.shape == g["image"].shape
    assert np.max(np.abs(img - g["image"])) <= 1e-6


def test_scalar_loop_equals_vector_form():
    g = load_golden("small_80x64_n120_tile8")
    pre = golden_preprocessed(g)
    a = cpu_ref.render_image(pre, int(g["width"]), int(g["height"]), int(g["tile"]), scalar=True)
    b = cpu_ref.render_image(pre, int(g["width"]), int(g["height"]), int(g["tile"]))
    assert np.array_equal(a, b)
    assert np.max(np.abs(a - g["image"])) <= 1e-6


def test_c_blend_given_reference_stage1(golden):
    g = golden
    img, pairs, inst = c_oracle.render(golden_preprocessed(g), int(g["width"]), int(g["height"]), int(g["tile"]))
    assert np.max(np.abs(img - g["image"])) <= 1e-6
    t = int(g["tile"])
    assert pairs == inst * t * t


def test_c_full_path_matches_reference(golden):
    g = golden
    pre = c_oracle.preprocess(g["points"], g["colors"], g["scales"], g["quaternions"], g["opacity"], oracle_camera(g))
    img, _, _ = c_oracle.render(pre, int(g["width"]), int(g["height"]), int(g["tile"]))
    assert np.max(np.abs(img - g["image"])) <= 1e-5


def test_reference_quirks_are_reproduced():
    g = load_golden("c1_256x256_n2000")
    img = g["image"]
    t = int(g["tile"])
    # last tile row/column never rendered (range(0, W - tile, tile))
    assert np.all(img[256 - t:, :, :] == 0) and np.all(img[:, 256 - t:, :] == 0)
    assert img[: 256 - t, : 256 - t].max() > 0.1
    # double sigmoid caps a single splat at sigmoid(1) * colour < 0.7311
    one = cpu_ref.render_pixel_scalar(
        5, 5, np.array([[5.0, 5.0]], np.float32), np.array([[1.0, 1.0, 1.0]], np.float32),
        np.array([1.0], np.float32), np.array([[[1.0, 0.0], [0.0, 1.0]]], np.float32))
    assert abs(one[0] - 1.0 / (1.0 + np.exp(-1.0))) < 1e-6


def test_c_window_and_thread_count_do_not_change_pixels():
    g = load_golden("c1_256x256_n2000")
    pre = golden_preprocessed(g)
    full, _, _ = c_oracle.render(pre, 256, 256, 16, nthreads=1)
    multi, _, _ = c_oracle.render(pre, 256, 256, 16, nthreads=4)
    assert np.array_equal(full, multi)
    win, pairs, _ = c_oracle.render(pre, 256, 256, 16, window=(2, 5, 3, 7))
    assert np.array_equal(win[32:80, 48:112], full[32:80, 48:112])
    assert np.all(win[:32] == 0) and np.all(win[80:] == 0) and pairs > 0


def test_empty_and_all_culled_scenes():
    g = load_golden("small_64x48_n300")
    cam = oracle_camera(g)
    z = np.zeros((0, 3), np.float32)
    pre = c_oracle.preprocess(z, z, z, np.zeros((0, 4), np.float32), np.zeros((0, 1), np.float32), cam)
    img, pairs, inst = c_oracle.render(pre, 64, 48, 16)
    assert pre.points.shape[0] == 0 and not img.any() and pairs == 0 and inst == 0
    # move everything behind the camera: nothing survives the z >= 0.2 cull
    R = np.asarray(g["world2view"])[:3, :3]
    behind = (np.array([[0.0, 0.0, -5.0]], np.float32) - np.asarray(g["world2view"])[3, :3]) @ R.T
    pts = np.repeat(behind.astype(np.float32), 7, axis=0)
    pre = cpu_ref.preprocess(pts, g["colors"][:7], g["scales"][:7], g["quaternions"][:7], g["opacity"][:7], cam)
    assert pre.points.shape[0] == 0


def test_cuda_semantics_restatement_against_numpy():
    """oracle/raster_cpu.c::orc_render_cuda_semantics vs an independent numpy statement of
    splat/c/render.cu:21-87 (unpinned by the reference itself: its kernel cannot run here)."""
    g = load_golden("small_64x48_n300")
    pre = golden_preprocessed(g)
    w, h = int(g["width"]), int(g["height"])
    img = c_oracle.render_cuda_semantics(pre, w, h, nthreads=2)
    f32 = np.float32
    PX, PY = np.meshgrid(np.arange(w), np.arange(h))           # (h, w)
    T = np.ones((h, w), f32)
    C = np.zeros((h, w, 3), f32)
    live = np.ones((h, w), bool)
    op = pre.sigmoid_opacity.reshape(-1)
    for i in range(pre.points.shape[0]):
        inside = (PX >= pre.min_x[i]) & (PX <= pre.max_x[i]) & (PY >= pre.min_y[i]) & (PY <= pre.max_y[i]) & live
        dx = (PX - int(pre.points[i, 0])).astype(f32)
        dy = (PY - int(pre.points[i, 1])).astype(f32)
        q = pre.inverse_covariance_2d[i]
        power = dx * q[0, 0] * dx + f32(2) * dx * dy * q[0, 1] + dy * dy * q[1, 1]
        alpha = np.minimum(f32(0.99), op[i] * np.exp(f32(-0.5) * power))
        test = T * (f32(1) - alpha)
        stop = inside & (test < f32(0.001))
        live &= ~stop
        acc = inside & ~stop
        C[acc] += (T * alpha)[acc][:, None] * pre.colors[i][None, :]
        T[acc] = test[acc]
    assert img.shape == (h, w, 3)
    assert np.max(np.abs(img - C)) <= 2e-6
    # differs from the CPU path by construction (single sigmoid, clamp, per-pixel cull, edge tiles)
    cpu_sem, _, _ = c_oracle.render(pre, w, h, 16)
    assert np.max(np.abs(img.transpose(1, 0, 2) - cpu_sem)) > 0.05


def test_dense_fixture_exercises_the_stop_rule():
    """dense_48x48_n1500 exists so that `T(1-alpha) < 1e-6 -> return before accumulating`
    (splat/gaussian_scene.py:166) is covered by a vector of the reference itself: in its first tile
    the rule must fire for a good share of the pixels (and the image parity tests then pin it)."""
    from oracle import cpu_ref

    g = load_golden("dense_48x48_n1500")
    pre = golden_preprocessed(g)
    lst = cpu_ref.tile_list(pre, 0, 0, 16)
    op2 = cpu_ref.sigmoid(pre.sigmoid_opacity[lst, 0]).astype(np.float64)
    pts, Q = pre.points[lst].astype(np.float64), pre.inverse_covariance_2d[lst].astype(np.float64)
    fired = 0
    for px in range(0, 16, 3):
        for py in range(0, 16, 3):
            d = pts - np.array([px, py], np.float64)
            w = np.exp(-0.5 * np.einsum("ni,nij,nj->n", d, Q, d)) * op2
            T = 1.0
            for a in w:
                if T * (1 - a) < 1e-6:
                    fired += 1
                    break
                T *= 1 - a
    assert fired >= 5
