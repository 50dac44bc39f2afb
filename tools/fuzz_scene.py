"""The scene of one seed of tools/fuzz.py (shared with tools/attic/fuzz_case.py and fuzz_bisect.py, which localise a seed
that failed): frame, tile size, population, pose, footprints.  Returns the random stream as well -- fuzz.py keeps drawing
from it (layout, window, ...)."""
import numpy as np

from intro_to_gaussian_splatting_amd.synthetic import make_scene


def fuzz_scene(seed: int, big: bool = False, extreme: bool = False, family: str = ""):
    if family:          # the generators behind the reference-made fixtures, at random sizes and poses
        from intro_to_gaussian_splatting_amd.synthetic import (make_few_visible_scene, make_needle_scene, make_tie_scene,
                                                                 make_trained_like_scene)
        rs = np.random.RandomState(99000 + seed)
        w, h = int(rs.randint(40, 700)), int(rs.randint(40, 500))
        tile = int(rs.choice([4, 8, 16, 16, 16, 32]))
        q = rs.normal(size=4) * 0.15 + np.array([0.96282662, -0.23562335, 0.12748722, 0.0345476])
        pose = dict(qvec=tuple(q / np.linalg.norm(q)), tvec=tuple(np.array([0.0530637, 0.87330016, 3.58750122]) + rs.normal(size=3) * 0.3))
        which = family if family != "mixed" else str(rs.choice(["trained", "needle", "tie", "few"]))
        if which == "trained":
            n = int(rs.choice([300, 3000, 30000]))
            sc = make_trained_like_scene(n, w, h, seed=seed, sh_degree=0, **pose)
        elif which == "needle":
            n = int(rs.choice([50, 400, 3000]))
            lo = float(rs.uniform(5.0, 60.0))
            sc = make_needle_scene(n, w, h, seed=seed, long_px=(lo, lo * float(rs.uniform(1.5, 4.0))),
                                   short_px=(float(rs.uniform(0.03, 0.3)), float(rs.uniform(0.3, 0.8))), **pose)
        elif which == "tie":
            n = int(rs.choice([100, 2000, 20000]))
            sc = make_tie_scene(n, w, h, seed=seed, levels=int(rs.choice([3, 33, 400])))
        else:
            n = int(rs.choice([3, 9, 40]))
            sc = make_few_visible_scene(n, w, h, seed=seed, visible=int(rs.randint(0, 4)))
        return rs, sc, w, h, tile, int(sc["points"].shape[0]), True
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
