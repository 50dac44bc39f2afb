"""Which Gaussians of a fuzz seed's scene does the stage-2 entry composite differently from the C restatement?  Bisects
over ranges of the depth order (the oracle's stage-1 arrays go through gsx_render_preprocessed).
    [GSX_TEST_LIB_PATH=...] python tools/attic/fuzz_bisect.py <seed>"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from intro_to_gaussian_splatting_amd import GaussianScene, Gaussians, _ffi, render_preprocessed
if os.environ.get("GSX_TEST_LIB_PATH"):
    _ffi.use_test_library()
from intro_to_gaussian_splatting_amd.synthetic import make_scene, write_colmap_text
from oracle import c_oracle, cpu_ref
seed = int(sys.argv[1])
rs = np.random.RandomState(77000 + seed)
w, h = int(rs.randint(3, 400)), int(rs.randint(3, 300)); tile = int(rs.choice([1, 2, 3, 4, 7, 8, 16, 16, 16, 16, 17, 32, 40])); n = int(rs.choice([0, 1, 2, 17, 300, 2500, 20000]))
w, h = int(rs.randint(300, 2200)), int(rs.randint(200, 1300)); tile = int(rs.choice([3, 4, 8, 16, 16, 16])); n = int(rs.choice([5000, 50000, 200000]))
q = rs.normal(size=4)
sc = make_scene(max(n, 1), w, h, seed=seed, behind_fraction=float(rs.choice([0.0, 0.0, 0.3, 1.0])), qvec=tuple(q / np.linalg.norm(q)), tvec=tuple(rs.normal(size=3)),
                spread=float(rs.choice([1.0, 1.0, 1.5, 3.0])), sigma_scale=float(rs.choice([0.05, 0.5, 1.0, 1.0, 3.0, 12.0])))
if rs.uniform() < 0.4 and n > 0:
    sc["scales"] = sc["scales"].copy()
    pick = rs.uniform(size=sc["scales"].shape[0]) < float(rs.choice([0.02, 0.2, 1.0]))
    sc["scales"][pick, rs.randint(0, 3)] *= np.float32(rs.uniform(20.0, 300.0))
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
