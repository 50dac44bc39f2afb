"""One seed of tools/fuzz.py's big profile through several kernel routes, to localise a mismatch.
    python tools/attic/fuzz_case.py <seed> [big] [extreme]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from intro_to_gaussian_splatting_amd import GaussianScene, Gaussians, _ffi
if os.environ.get("GSX_TEST_LIB_PATH") or os.environ.get("GSX_FUZZ_TEST_LIB"):
    _ffi.use_test_library()
from intro_to_gaussian_splatting_amd.synthetic import make_scene, write_colmap_text
from oracle import c_oracle, cpu_ref
seed = int(sys.argv[1])
from tools.fuzz_scene import fuzz_scene
_, sc, w, h, tile, n, needles = fuzz_scene(seed, "big" in sys.argv[2:], "extreme" in sys.argv[2:])
print("seed", seed, "w h tile n", w, h, tile, n, "needles", needles)
with tempfile.TemporaryDirectory() as tmp:
    write_colmap_text(tmp, sc)
    g = Gaussians.from_arrays(sc["points"], sc["colors_0_255"], sc["scales"], sc["quaternions"], sc["opacity"], device="cuda:0")
    scene = GaussianScene(tmp, g)
im = scene.images[1]
c = im.gsx_camera()
cam = cpu_ref.Camera(im.world2view.cpu().numpy(), im.full_proj_transform.cpu().numpy(), np.float32(c.tan_fovx),
                     np.float32(c.tan_fovy), np.float32(c.fx), np.float32(c.fy), c.width, c.height)
pre = c_oracle.preprocess(sc["points"], g.colors.cpu().numpy(), sc["scales"], sc["quaternions"], sc["opacity"], cam)
ref, _, inst = c_oracle.render(pre, w, h, tile)
def show(tag, img, st):
    img = img.cpu().numpy()
    d = np.abs(img - ref).max(axis=-1)
    ix = np.unravel_index(np.argmax(d), d.shape)
    print("%-34s max %.3g at pixel %s (tile %s), pixels > 1e-4: %d, stats %s" % (
        tag, d.max(), ix, (ix[0] // tile, ix[1] // tile), int((d > 1e-4).sum()), {k: st.get(k) for k in ("n_instances", "n_redo", "plain_footprints")}))
    return img
for tag, kw in (("default", {}), ("again (hints)", {}), ("no long-tile split", dict(split_long_tiles=False)),
                ("generic kernels", dict(generic_kernels=True)), ("no hints", dict(use_hints=False)),
                ("no schedule", dict(tile_schedule=False))):
    st = {}
    show(tag, scene.render_image_hip(1, tile_size=tile, stats=st, **kw), st)
print("oracle instances", inst)
