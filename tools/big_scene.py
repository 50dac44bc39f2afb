"""Sizes beyond the BASELINE configurations: N Gaussians (default 32M) at 1920x1080 through the whole HIP path, against
the C restatement (oracle/raster_cpu.c) -- counts equal, pixels within 1e-4 -- and as eight tile-column strips (== the frame).
    python tools/big_scene.py [N] [width] [height]
A 288 GB part holds scenes of hundreds of millions of Gaussians; this is the check that nothing on the path counts in 24 or
31 bits where it should not."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from intro_to_gaussian_splatting_amd import GaussianScene, Gaussians  # noqa: E402
from intro_to_gaussian_splatting_amd.synthetic import make_scene, write_colmap_text  # noqa: E402
from oracle import c_oracle  # noqa: E402
import tempfile  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 32_000_000
w = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
h = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
t0 = time.time()
# footprints shrink with the population so that the pair count stays near C4's (sigma_scale ~ 1 / sqrt(n / 1M))
sc = make_scene(n, w, h, seed=0, sigma_scale=float(min(1.0, (1e6 / n) ** 0.5 * 2.0)))
print("scene of %d Gaussians generated in %.1f s" % (n, time.time() - t0), flush=True)
with tempfile.TemporaryDirectory() as tmp:
    write_colmap_text(tmp, sc)
    g = Gaussians.from_arrays(sc["points"], sc["colors_0_255"], sc["scales"], sc["quaternions"], sc["opacity"], device="cuda:0")
    scene = GaussianScene(tmp, g)
st = {}
img = scene.render_image_hip(1, stats=st)
torch.cuda.synchronize()
for _ in range(3):
    scene.render_image_hip(1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    scene.render_image_hip(1)
e1.record()
torch.cuda.synchronize()
print("HIP: visible %d, kept %d, pairs %d; %.3f ms per frame (separate launches, hinted)" % (
    st["n_visible"], st.get("n_kept", -1), st["n_instances"], e0.elapsed_time(e1) / 5), flush=True)
t0 = time.time()
cam = bench._oracle_camera(scene, 1)
pre = c_oracle.preprocess(sc["points"], scene.gaussians.colors.cpu().numpy(), sc["scales"], sc["quaternions"], sc["opacity"], cam)
port, _, inst = c_oracle.render(pre, w, h, 16)
print("C port: visible %d, pairs %d in %.1f s" % (pre.points.shape[0], inst, time.time() - t0), flush=True)
d = np.abs(img.cpu().numpy().astype(np.float64) - port).max()
print("max |dpixel| vs the C port %.3g" % d)
assert st["n_visible"] == pre.points.shape[0] and st["n_instances"] == inst and d <= 1e-4
# the same frame as eight strips of tile columns
ntx = (w + 15) // 16
parts = torch.empty_like(img)
for k in range(8):
    c0, c1 = ntx * k // 8, ntx * (k + 1) // 8
    scene.render_image_hip(1, tile_window=(c0, c1, 0, (h + 15) // 16), out=parts[c0 * 16:min(c1 * 16, w)], out_origin=(c0 * 16, 0))
torch.cuda.synchronize()
assert torch.equal(parts, img), "strips != frame"
print("eight strips == the frame, bit for bit; OK")
