"""GSX_SEM_STD_3DGS oracle (build extension, PARITY UNPINNED -- no reference code for it exists in
/root/reference): the scalar C restatement (oracle/raster_cpu.c:orc_render_std3dgs) and the
vectorised numpy restatement (oracle/std3dgs_ref.py) of the published 3DGS forward pass must agree,
and known-answer cases derived by hand from the published rules must hold."""
import numpy as np
import pytest

from intro_to_gaussian_splatting_amd import synthetic
from oracle import c_oracle, cpu_ref, std3dgs_ref


def _scene(n, w, h, seed, behind=0.0):
    sc = synthetic.make_scene(n, w, h, seed=seed, behind_fraction=behind)
    cam = cpu_ref.build_camera(sc["qvec"], sc["tvec"], sc["fx"], sc["fy"], w, h)
    colors = (sc["colors_0_255"] / np.float32(256.0)).astype(np.float32)
    return sc, cam, colors


def _outliers(a, b, tol=1e-4):
    d = np.abs(a.astype(np.float64) - b.astype(np.float64)).max(axis=-1)
    return d, int((d > tol).sum())


@pytest.mark.parametrize("n,w,h,tile,seed,behind", [(300, 64, 48, 16, 0, 0.0), (500, 100, 70, 16, 1, 0.2),
                                                     (200, 50, 50, 8, 2, 0.0), (400, 96, 80, 32, 3, 0.1)])
def test_c_and_numpy_restatements_agree(n, w, h, tile, seed, behind):
    sc, cam, colors = _scene(n, w, h, seed, behind)
    bg = (0.1, 0.2, 0.3)
    img_c, nvis_c, inst_c, st_c = c_oracle.render_std3dgs(sc["points"], colors, sc["scales"], sc["quaternions"],
                                                          sc["opacity"], cam, tile=tile, background=bg, nthreads=2)
    img_n, nvis_n, inst_n = std3dgs_ref.render(sc["points"], colors, sc["scales"], sc["quaternions"],
                                               sc["opacity"], cam, tile=tile, background=bg)
    assert nvis_c == nvis_n and inst_c == inst_n
    # stage 1: same float32 operation order -> bit-identical where the C code filled a row
    st = std3dgs_ref.stage1(sc["points"], sc["scales"], sc["quaternions"], sc["opacity"], cam, tile)
    filled = ~np.isnan(st_c[:, 0])
    assert filled.sum() >= st.keep.sum()
    np.testing.assert_array_equal(st_c[filled, 0:2], st.xy[filled])
    np.testing.assert_array_equal(st_c[filled, 2:5], st.conic[filled])
    np.testing.assert_array_equal(st_c[filled, 5], st.radius[filled])
    np.testing.assert_array_equal(st_c[filled, 6], st.depth[filled])
    # stage 2: libm expf vs numpy exp differ by an ulp; a pixel may flip at the 1/255 threshold
    d, bad = _outliers(img_c, img_n)
    assert bad <= 2 and d.max() < 0.006, (bad, d.max())


def test_window_renders_only_its_tiles():
    sc, cam, colors = _scene(300, 64, 48, 4)
    full, _, inst_full, _ = c_oracle.render_std3dgs(sc["points"], colors, sc["scales"], sc["quaternions"],
                                                    sc["opacity"], cam, nthreads=1)
    part, _, inst_part, _ = c_oracle.render_std3dgs(sc["points"], colors, sc["scales"], sc["quaternions"],
                                                    sc["opacity"], cam, nthreads=1, window=(1, 3, 0, 2))
    np.testing.assert_array_equal(part[0:32, 16:48], full[0:32, 16:48])
    assert not part[32:].any() and not part[:, :16].any() and not part[:, 48:].any()
    assert 0 < inst_part < inst_full


def test_known_answers_single_gaussian():
    """One isotropic Gaussian on the optical axis, identity pose: every number below follows from
    the published rules by hand."""
    w = h = 32
    fx = fy = 16.0                                   # tan(fov/2) = 1, focal used by the rules = W/2 = 16
    cam = cpu_ref.build_camera((1.0, 0.0, 0.0, 0.0), (0.0, 0.0, 0.0), fx, fy, w, h)
    pts = np.array([[0.0, 0.0, 4.0]], np.float32)
    scales = np.full((1, 3), 0.5, np.float32)        # sigma_px = 0.5 * 16 / 4 = 2  -> cov = 4 + 0.3
    quats = np.array([[1.0, 0.0, 0.0, 0.0]], np.float32)
    logit = np.array([[0.0]], np.float32)            # opacity 0.5
    colors = np.array([[1.0, 0.5, 0.25]], np.float32)
    bg = (0.0, 0.0, 1.0)
    img, nvis, inst, st = c_oracle.render_std3dgs(pts, colors, scales, quats, logit, cam, background=bg, nthreads=1)
    assert nvis == 1
    x, y, A, B, C, r, depth, op = st[0]
    assert abs(x - 15.5) < 1e-4 and abs(y - 15.5) < 1e-4          # ((0 + 1) * 32 - 1) / 2
    assert abs(A - 1 / 4.3) < 1e-5 and abs(C - 1 / 4.3) < 1e-5 and abs(B) < 1e-6
    assert r == np.ceil(3 * np.sqrt(4.3 + np.sqrt(0.1))) and depth == 4.0 and abs(op - 0.5) < 1e-7
    assert inst == 4                                               # [15.5-8, 15.5+8] touches 2 x 2 tiles of 16
    # pixel (15,15): d = 0.5, 0.5 -> power = -0.5 * 0.5 / 4.3
    alpha = 0.5 * np.exp(-0.5 * (0.25 + 0.25) / 4.3)
    np.testing.assert_allclose(img[15, 15], np.array([1.0, 0.5, 0.25]) * alpha + (1 - alpha) * np.array(bg), atol=1e-6)
    # far pixel: alpha < 1/255 -> skipped, pure background
    np.testing.assert_array_equal(img[0, 0], np.array(bg, np.float32))
    # same through the numpy restatement
    img_n, _, _ = std3dgs_ref.render(pts, colors, scales, quats, logit, cam, background=bg)
    np.testing.assert_allclose(img_n, img, atol=1e-6)


def _stack(n, logit_value, z0=2.0):
    w = h = 16
    cam = cpu_ref.build_camera((1.0, 0.0, 0.0, 0.0), (0.0, 0.0, 0.0), 8.0, 8.0, w, h)
    pts = np.tile(np.array([[0.0, 0.0, z0]], np.float32), (n, 1))
    pts[:, 2] += np.arange(n, dtype=np.float32) * 0.01
    scales = np.full((n, 3), 2.0, np.float32)        # huge footprint: exp(power) ~ 1 at the centre
    quats = np.tile(np.array([[1.0, 0.0, 0.0, 0.0]], np.float32), (n, 1))
    logit = np.full((n, 1), logit_value, np.float32)
    return cam, pts, scales, quats, logit


def test_alpha_clamp_and_stop_rule():
    """Opaque coincident Gaussians: alpha clamps to 0.99; in float32 T(1 - alpha) for the second one is
    0.01f * (1 - 0.99f) = 9.99998e-5 < 1e-4, so the pixel stops after ONE Gaussian."""
    cam, pts, scales, quats, logit = _stack(10, 20.0)
    colors = np.zeros((10, 3), np.float32)
    colors[0] = (1.0, 0.0, 0.0)
    colors[1] = (0.0, 1.0, 0.0)
    img, _, _, _ = c_oracle.render_std3dgs(pts, colors, scales, quats, logit, cam, background=(0, 0, 1), nthreads=1)
    np.testing.assert_allclose(img[8, 8], [0.99, 0.0, 0.01], rtol=1e-5, atol=1e-7)   # + T * background
    img_n, _, _ = std3dgs_ref.render(pts, colors, scales, quats, logit, cam, background=(0, 0, 1))
    np.testing.assert_allclose(img_n[8, 8], img[8, 8], atol=1e-7)


def test_stop_rule_counts_gaussians():
    """alpha ~ 0.5 each: T halves per Gaussian; the 14th would leave T = 2^-14 = 6.1e-5 < 1e-4, so
    exactly 13 are composited: C = 1 - 2^-13 (up to the footprint factor), T_final = 2^-13."""
    cam, pts, scales, quats, logit = _stack(20, 0.0)
    colors = np.ones((20, 3), np.float32)
    img, _, _, st = c_oracle.render_std3dgs(pts, colors, scales, quats, logit, cam, nthreads=1)
    # exact expectation from the stage-1 numbers, in float64
    px = np.array([8.0, 8.0])
    T, C, count = 1.0, 0.0, 0
    for x, y, A, B, Cc, r, depth, op in st.astype(np.float64):
        dx, dy = x - px[0], y - px[1]
        alpha = min(0.99, op * np.exp(-0.5 * (A * dx * dx + Cc * dy * dy) - B * dx * dy))
        if T * (1 - alpha) < 1e-4:
            break
        C += alpha * T
        T *= 1 - alpha
        count += 1
    assert count == 13
    np.testing.assert_allclose(img[8, 8], [C, C, C], rtol=1e-5)
