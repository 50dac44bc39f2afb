"""The reference's own GPU configuration (render.cu, 52k Gaussians, ~16.9 Mpixel
frame, 2.48 s on an sm_89 GPU) through this library with the same rules (semantics="ref_cuda"):
render_image_cuda = preprocess + the native entry point on its arrays."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tempfile
import torch
from intro_to_gaussian_splatting_amd import GaussianScene, Gaussians
from intro_to_gaussian_splatting_amd.synthetic import make_scene, write_colmap_text

n, w, h = 52363, 5187, 3361
sc = make_scene(n, w, h, seed=0)
with tempfile.TemporaryDirectory() as tmp:
    write_colmap_text(tmp, sc)
    g = Gaussians.from_arrays(sc["points"], sc["colors_0_255"], sc["scales"], sc["quaternions"], sc["opacity"], device="cuda:0")
    scene = GaussianScene(tmp, g)
for name, fn in (("render_image_cuda (ref_cuda rules, preprocess + stage 2)", lambda: scene.render_image_cuda(1)),
                 ("render_image_hip ref_cuda (whole path)", lambda: scene.render_image_hip(1, layout="hw3", semantics="ref_cuda")),
                 ("render_image_hip ref_cpu (whole path)", lambda: scene.render_image_hip(1))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print("%s: %.2f ms/frame (%dx%d = %.1f Mpixel, %d Gaussians) -> %.0f Mpixel/s" % (name, dt * 1e3, w, h, w * h / 1e6, n, w * h / dt / 1e6))
