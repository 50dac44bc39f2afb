"""Writes a synthetic scene in the trained-3DGS .ply layout (for `bench.py --ply`, which otherwise
needs a user-supplied checkpoint): the section-8(d) generator's Gaussians, colour as SH of the given
degree (DC from the generator's rgb, higher bands small random).

    python tools/make_synthetic_ply.py out.ply [n] [sh_degree] [width] [height]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intro_to_gaussian_splatting_amd import ply  # noqa: E402
from intro_to_gaussian_splatting_amd.synthetic import make_scene  # noqa: E402

out = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
deg = int(sys.argv[3]) if len(sys.argv) > 3 else 3
w = int(sys.argv[4]) if len(sys.argv) > 4 else 1920
h = int(sys.argv[5]) if len(sys.argv) > 5 else 1080
sc = make_scene(n, w, h, seed=0)
k = (deg + 1) ** 2
rs = np.random.RandomState(1)
sh = np.zeros((n, k, 3), np.float32)
sh[:, 0] = (sc["colors_0_255"] / 256.0 - 0.5) / 0.28209479177387814      # colour = 0.5 + C0 * dc
sh[:, 1:] = rs.normal(0.0, 0.05, size=(n, k - 1, 3)).astype(np.float32)
ply.save_trained(out, sc["points"], sh, sc["scales"], sc["quaternions"], sc["opacity"])
print("wrote %s: %d Gaussians, SH degree %d (%.1f MB)" % (out, n, deg, os.path.getsize(out) / 1e6))
