"""The scene of one seed of tools/fuzz.py (shared with tools/attic/fuzz_case.py and fuzz_bisect.py, which localise a seed
that failed): frame, tile size, population, pose, footprints.  Returns the random stream as well -- fuzz.py keeps drawing
from it (layout, window, ...)."""
import numpy as np

from intro_to_gaussian_splatting_amd.synthetic import make_scene


def fuzz_scene(seed: int, big: bool = False, extreme: bool = False):
    rs = np.random.RandomState(77000 + seed)
    w, h = int(rs.randint(3, 400)), int(rs.randint(3, 300))
    tile = int(rs.choice([1, 2, 3, 4, 7, 8, 16, 16, 16, 16, 17, 32, 40]))
    n = int(rs.choice([0, 1, 2, 17, 300, 2500, 20000]))
    if big:
        w, h = int(rs.randint(300, 2200)), int(rs.randint(200, 1300))
        tile = int(rs.choice([3, 4, 8, 16, 16, 16]))
        n = int(rs.choice([5000, 50000, 200000]))
    q = rs.normal(size=4)
    sc = make_scene(max(n, 1), w, h, seed=seed, behind_fraction=float(rs.choice([0.0, 0.0, 0.3, 1.0])),
                    qvec=tuple(q / np.linalg.norm(q)), tvec=tuple(rs.normal(size=3)),
                    spread=float(rs.choice([1.0, 1.0, 1.5, 3.0])), sigma_scale=float(rs.choice([0.05, 0.5, 1.0, 1.0, 3.0, 12.0])))
    needles = False
    if rs.uniform() < 0.4 and n > 0:
        needles = True
        # needles: a share of the Gaussians stretched 20 .. 300-fold along one axis -- ill-conditioned footprints, which
        # the compositing kernels evaluate in the reference's operation order (gsx_blend.hip: kKindRefOrder)
        sc["scales"] = sc["scales"].copy()
        pick = rs.uniform(size=sc["scales"].shape[0]) < float(rs.choice([0.02, 0.2, 1.0]))
        sc["scales"][pick, rs.randint(0, 3)] *= np.float32(rs.uniform(20.0, 300.0))
    if extreme and n > 0:
        # the corners of the parameter space: needles to 3000:1, pancakes (two axes stretched), specks (sigma of a hundredth
        # of a pixel), opacity logits of +-12, everything on one spot
        needles = True
        sc["scales"] = sc["scales"].copy()
        sc["opacity"] = sc["opacity"].copy()
        m_ = sc["scales"].shape[0]
        kind = rs.randint(0, 5, size=m_)
        ax = rs.randint(0, 3, size=m_)
        sc["scales"][np.arange(m_)[kind == 0], ax[kind == 0]] *= rs.uniform(300.0, 3000.0, int((kind == 0).sum())).astype(np.float32)
        for a_ in (0, 1):
            sel = kind == 1
            sc["scales"][np.arange(m_)[sel], (ax[sel] + a_) % 3] *= rs.uniform(30.0, 400.0, int(sel.sum())).astype(np.float32)
        sc["scales"][kind == 2] *= np.float32(0.01)
        sc["opacity"][kind == 3] = np.where(rs.uniform(size=(int((kind == 3).sum()), 1)) < 0.5, 12.0, -12.0).astype(np.float32)
        if rs.uniform() < 0.3:
            sc["points"] = sc["points"].copy()
            sc["points"][kind == 4] = sc["points"][0]
    return rs, sc, w, h, tile, n, needles
