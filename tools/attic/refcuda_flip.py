"""A tools/fuzz.py seed whose ref_cuda frame (the reference CUDA kernel's rules; GSX_SEM_REF_CUDA) is further from the C
restatement than the fuzz's heuristic bar: is it a flip of the stop test T (1 - alpha) < 1e-3 (render.cu:72-76)?  Walks the
worst pixel in float32 with the restatement's operations and prints the records whose test lies near the threshold.
    python tools/attic/refcuda_flip.py <seed>"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from intro_to_gaussian_splatting_amd import GaussianScene, Gaussians
from intro_to_gaussian_splatting_amd.synthetic import write_colmap_text
from oracle import c_oracle, cpu_ref
from tools.fuzz_scene import fuzz_scene

seed = int(sys.argv[1])
rs, sc, w, h, tile, n, needles = fuzz_scene(seed, False, False, "")
with tempfile.TemporaryDirectory() as tmp:
    write_colmap_text(tmp, sc)
    g = Gaussians.from_arrays(sc["points"], sc["colors_0_255"], sc["scales"], sc["quaternions"], sc["opacity"], device="cuda:0")
    scene = GaussianScene(tmp, g)
im = scene.images[1]
c = im.gsx_camera()
cam = cpu_ref.Camera(im.world2view.cpu().numpy(), im.full_proj_transform.cpu().numpy(), np.float32(c.tan_fovx), np.float32(c.tan_fovy),
                     np.float32(c.fx), np.float32(c.fy), c.width, c.height)
pre = c_oracle.preprocess(sc["points"], g.colors.cpu().numpy(), sc["scales"], sc["quaternions"], sc["opacity"], cam)
cref = c_oracle.render_cuda_semantics(pre, w, h)
cimg = scene.render_image_hip(1, tile_size=tile, layout="hw3", semantics="ref_cuda").cpu().numpy()
dc = np.abs(cimg.astype(np.float64) - cref).max(axis=-1)
py, px = np.unravel_index(int(np.argmax(dc)), dc.shape)
print("seed %d: %dx%d tile %d n %d; pixels above 1e-4: %d, worst %.3e at (x %d, y %d): kernel %s restatement %s" % (
    seed, w, h, tile, n, int((dc > 1e-4).sum()), dc.max(), px, py, cimg[py, px], cref[py, px]))
f32 = np.float32
T = f32(1.0)
col = np.zeros(3, f32)
for i in range(len(pre.depths)):
    if not (f32(py) >= pre.min_y[i] and f32(py) <= pre.max_y[i] and f32(px) >= pre.min_x[i] and f32(px) <= pre.max_x[i]):
        continue
    ix, iy = int(pre.points_xy[i, 0]), int(pre.points_xy[i, 1])
    dx, dy = f32(px - ix), f32(py - iy)
    q = pre.inverse_covariance_2d[i].reshape(-1)
    power = f32(f32(f32(dx * q[0]) * dx) + f32(f32(f32(f32(2) * dx) * dy) * q[1])) + f32(f32(dy * dy) * q[3])
    alpha = min(f32(0.99), f32(pre.sigmoid_opacity.reshape(-1)[i] * np.exp(f32(-0.5) * power, dtype=f32)))
    test = f32(T * f32(f32(1) - alpha))
    near = abs(float(test) - 1e-3) < 2e-6
    if near or test < f32(0.001):
        print("  record %d: T %.9g alpha %.6f test %.9g (threshold 0.001, %+.2e away)  -> T alpha = %.4e: what a flip of this test moves the pixel by (x colour %s)" % (
            i, T, alpha, test, float(test) - 1e-3, float(T * alpha), pre.colors[i]))
    if test < f32(0.001):
        break
    col = col + f32(T * alpha) * pre.colors[i]
    T = test
print("  float32 walk of the restatement's rules: %s" % col)
