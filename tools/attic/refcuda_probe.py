"""The reference's own notebook workload (Treehill image 100, 52 363 Gaussians from
the constructor defaults -- scale 0.001, identity rotation, opacity 0.9999 -- native 5068x3328, tile 16;
CPU render_image 343 s, CUDA render_image_cuda 2.4787 s on an sm_89 GPU) on a synthetic stand-in of the
same size (points uniform in the frustum; the Treehill point cloud is not available offline), with the
reference's two rule sets."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tempfile
import torch
from intro_to_gaussian_splatting_amd import GaussianScene, Gaussians
from intro_to_gaussian_splatting_amd.synthetic import make_scene, write_colmap_text

n, w, h = 52363, 5068, 3328
sc = make_scene(n, w, h, seed=0)
with tempfile.TemporaryDirectory() as tmp:
    write_colmap_text(tmp, sc)
    g = Gaussians(torch.from_numpy(sc["points"]), torch.from_numpy(sc["colors_0_255"]), device="cuda:0")   # defaults
    scene = GaussianScene(tmp, g)
st = {}
scene.render_image_hip(1, stats=st)
print("%d Gaussians, %dx%d = %.2f Mpixel, %d visible, %d (Gaussian, tile) pairs" % (n, w, h, w * h / 1e6, st["n_visible"], st["n_instances"]))
for name, fn, ref in (("render_image (CPU rules; device frame)", lambda: scene.render_image_hip(1), 343.0),
                      ("render_image (CPU rules; host tensor like the reference)", lambda: scene.render_image(1), 343.0),
                      ("render_image_cuda (CUDA-kernel rules: preprocess + native entry point)", lambda: scene.render_image_cuda(1), 2.4787)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print("%s: %.2f ms/frame -> %.0f Mpixel/s (reference notebook: %.4g s)" % (name, dt * 1e3, w * h / dt / 1e6, ref))
