"""Which Gaussians of a fuzz seed's scene does the stage-2 entry composite differently from the C restatement?  Bisects
over ranges of the depth order (the oracle's stage-1 arrays go through gsx_render_preprocessed).
    [GSX_TEST_LIB_PATH=...] python tools/attic/fuzz_bisect.py <seed> [big] [extreme]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from intro_to_gaussian_splatting_amd import GaussianScene, Gaussians, _ffi, render_preprocessed
if os.environ.get("GSX_TEST_LIB_PATH"):
    _ffi.use_test_library()
from intro_to_gaussian_splatting_amd.synthetic import make_scene, write_colmap_text
from oracle import c_oracle, cpu_ref
seed = int(sys.argv[1])
from tools.fuzz_scene import fuzz_scene
_, sc, w, h, tile, n, needles = fuzz_scene(seed, "big" in sys.argv[2:], "extreme" in sys.argv[2:])
with tempfile.TemporaryDirectory() as tmp:
    write_colmap_text(tmp, sc)
    g = Gaussians.from_arrays(sc["points"], sc["colors_0_255"], sc["scales"], sc["quaternions"], sc["opacity"], device="cuda:0")
    scene = GaussianScene(tmp, g)
im = scene.images[1]; c = im.gsx_camera()
cam = cpu_ref.Camera(im.world2view.cpu().numpy(), im.full_proj_transform.cpu().numpy(), np.float32(c.tan_fovx), np.float32(c.tan_fovy), np.float32(c.fx), np.float32(c.fy), c.width, c.height)
pre = c_oracle.preprocess(sc["points"], g.colors.cpu().numpy(), sc["scales"], sc["quaternions"], sc["opacity"], cam)
m = len(pre.depths)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
def diff(lo, hi):
    sub = pre._replace(**{k: np.ascontiguousarray(np.asarray(getattr(pre, k))[lo:hi]) for k in pre._fields if k != "order"})
    ref, _, _ = c_oracle.render(sub, w, h, tile)
    img = render_preprocessed(h, w, tile, t(sub.points), t(sub.colors), t(sub.inverse_covariance_2d), t(sub.min_x), t(sub.max_x), t(sub.min_y), t(sub.max_y), t(sub.sigmoid_opacity)).cpu().numpy()
    return float(np.abs(img - ref).max())
print("all", m, diff(0, m))
lo, hi = 0, m
while hi - lo > 1:
    mid = (lo + hi) // 2
    a, b = diff(lo, mid), diff(mid, hi)
    print("  [%d,%d) %.3g   [%d,%d) %.3g" % (lo, mid, a, mid, hi, b))
    if max(a, b) < 1e-4: break
    lo, hi = (lo, mid) if a >= b else (mid, hi)
if hi - lo == 1:
    i = lo
    Q = np.asarray(pre.inverse_covariance_2d)[i]
    print("culprit rank", i, "index", int(np.asarray(pre.order)[i]) if hasattr(pre, "order") else None, "Q", Q.ravel().tolist(), "cov", np.asarray(pre.covariance_2d)[i].ravel().tolist(),
          "xy", np.asarray(pre.points)[i].tolist(), "radius", float(np.asarray(pre.radius).reshape(-1)[i]), "op", float(np.asarray(pre.sigmoid_opacity).reshape(-1)[i]),
          "bbox", [float(np.asarray(getattr(pre, k)).reshape(-1)[i]) for k in ("min_x", "max_x", "min_y", "max_y")])
if "singles" in sys.argv:      # ... singles <lo> <hi>: every record of a range on its own
    k = sys.argv.index("singles")
    a_, b_ = int(sys.argv[k + 1]), int(sys.argv[k + 2])
    shown = 0
    for i in range(a_, b_):
        d = diff(i, i + 1)
        if d > 1e-4 and shown < 12:
            shown += 1
            Q = np.asarray(pre.inverse_covariance_2d)[i]
            print("rank", i, "alone %.3g" % d, "Q", Q.ravel().tolist(), "xy", np.asarray(pre.points)[i].tolist(),
                  "radius", float(np.asarray(pre.radius).reshape(-1)[i]), "op", float(np.asarray(pre.sigmoid_opacity).reshape(-1)[i]),
                  "cov", np.asarray(pre.covariance_2d)[i].ravel().tolist())
    print("singles done", a_, b_, "shown", shown)
if "walk" in sys.argv:      # ... walk <lo> <hi>: the worst pixel of a range, record by record, as the restatement composites it
    k = sys.argv.index("walk")
    a_, b_ = int(sys.argv[k + 1]), int(sys.argv[k + 2])
    sub = pre._replace(**{f: np.ascontiguousarray(np.asarray(getattr(pre, f))[a_:b_]) for f in pre._fields if f != "order"})
    ref, _, _ = c_oracle.render(sub, w, h, tile)
    img = render_preprocessed(h, w, tile, t(sub.points), t(sub.colors), t(sub.inverse_covariance_2d), t(sub.min_x), t(sub.max_x), t(sub.min_y), t(sub.max_y), t(sub.sigmoid_opacity)).cpu().numpy()
    d = np.abs(img - ref).max(axis=-1)
    px, py = np.unravel_index(np.argmax(d), d.shape)
    print("worst pixel", (px, py), "tile", (px // tile, py // tile), "gpu", img[px, py], "cpu", ref[px, py], "pixels off", int((d > 1e-4).sum()))
    f32 = np.float32
    T = f32(1.0)
    for i in range(b_ - a_):
        x, y = np.asarray(sub.points)[i]
        mnx, mxx, mny, mxy = [f32(np.asarray(getattr(sub, q_)).reshape(-1)[i]) for q_ in ("min_x", "max_x", "min_y", "max_y")]
        tx0, ty0 = (px // tile) * tile, (py // tile) * tile
        inside = (mnx <= tx0 + tile) and (mxx >= tx0) and (mny <= ty0 + tile) and (mxy >= ty0)
        Q = np.asarray(sub.inverse_covariance_2d)[i].astype(np.float64)
        e = np.array([float(x) - px, float(y) - py])
        power = -0.5 * e @ Q @ e
        op = 1.0 / (1.0 + np.exp(-float(np.asarray(sub.sigmoid_opacity).reshape(-1)[i])))
        alpha = np.exp(power) * op if power < 700 else np.inf
        print("  rec %3d in tile list %s  power(f64) %12.5g  alpha(f64) %10.4g  T before %.6g  Q %s radius %.0f" % (
            a_ + i, inside, power, alpha, float(T), np.asarray(sub.inverse_covariance_2d)[i].ravel().tolist(), float(np.asarray(sub.radius).reshape(-1)[i])))
        if inside:
            test = T * f32(1 - alpha) if np.isfinite(alpha) else -np.inf
            if test < 1e-6:
                print("     -> stops here")
                break
            T = f32(test)
