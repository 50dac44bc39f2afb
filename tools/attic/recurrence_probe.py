"""CPU study for the compositing loop's alpha recurrence.

alpha(y) = exp2(s0 - w(y)^2), w(y) = c - r11 y along the lane's four pixels y0 .. y0 + 3 is a geometric recurrence:
    a0 = exp2(s0 - w0^2), g0 = exp2(2 r11 w0 - r11^2), h = exp2(-2 r11^2)
    a1 = a0 g0, g1 = g0 h, a2 = a1 g1, g2 = g1 h, a3 = a2 g2
This script emulates both float32 forms (direct: what round 3 shipped; recurrence) in numpy on the records of a
synthetic scene, tile by tile, against float64, and prints the error of alpha for records the kernel would stage --
by class of r11 -- so that the rule that sends steep records to the direct form can be chosen on evidence.

    python tools/attic/recurrence_probe.py [--n 20000] [--clustered]
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from intro_to_gaussian_splatting_amd.synthetic import make_scene  # noqa: E402
from oracle import c_oracle, cpu_ref  # noqa: E402

f32, f64 = np.float32, np.float64


def fma(a, b, c):
    return (a.astype(f64) * b.astype(f64) + c.astype(f64)).astype(f32)


def exp2_32(x):
    with np.errstate(over="ignore", under="ignore"):
        r = np.exp2(x.astype(f64)).astype(f32)
    # v_exp_f32 flushes denormal results
    return np.where(np.abs(r) < np.float32(1.1754944e-38), f32(0), r)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=20000)
    ap.add_argument("--clustered", action="store_true")
    ap.add_argument("--gmax", type=float, default=64.0)
    ap.add_argument("--products", action="store_true", help="(a2, a3) = (a0, a1) * (g0^2 hx, g0^2 hx^3) instead of the chain")
    args = ap.parse_args()
    w, h = 1920, 1080
    kw = dict(cluster_fraction=0.5, cluster_area=0.05, sigma_ln=1.0) if args.clustered else {}
    sc = make_scene(args.n, w, h, seed=1, **kw)
    cam = cpu_ref.build_camera(sc["qvec"], sc["tvec"], float(sc["fx"]), float(sc["fy"]), w, h)
    pre = c_oracle.preprocess(sc["points"], sc["colors_0_255"] / 256.0, sc["scales"], sc["quaternions"], sc["opacity"], cam)
    q = pre.inverse_covariance_2d.astype(f64)
    kd = 0.5 * 1.44269504088896340736
    m00, m01, m11 = kd * q[:, 0, 0], kd * 0.5 * (q[:, 0, 1] + q[:, 1, 0]), kd * q[:, 1, 1]
    r11 = np.sqrt(m11)
    hh = m01 / r11
    d1 = m00 - hh * hh
    r11, hh, d1 = r11.astype(f32), hh.astype(f32), d1.astype(f32)
    sig = 1.0 / (1.0 + np.exp(-pre.sigmoid_opacity[:, 0].astype(f64)))
    lop = np.log2(sig).astype(f32)
    T = 16
    # (gaussian, tile) pairs by the reference's rule (gaussian_scene.py:209-217)
    gi, tx, ty = [], [], []
    for i in range(len(r11)):
        x_lo = max(0, int(np.ceil((pre.min_x[i] - T) / T)))
        x_hi = min((w - T - 1) // T, int(np.floor(pre.max_x[i] / T)))
        y_lo = max(0, int(np.ceil((pre.min_y[i] - T) / T)))
        y_hi = min((h - T - 1) // T, int(np.floor(pre.max_y[i] / T)))
        for a in range(x_lo, x_hi + 1):
            for b in range(y_lo, y_hi + 1):
                gi.append(i), tx.append(a), ty.append(b)
    gi, tx, ty = np.asarray(gi), np.asarray(tx, f32), np.asarray(ty, f32)
    P = len(gi)
    print("pairs", P, "per Gaussian %.2f" % (P / len(r11)))
    xr = (pre.points[gi, 0] - tx * T).astype(f32)      # tile-relative mean
    yr = (pre.points[gi, 1] - ty * T).astype(f32)
    R, H, D, L = r11[gi], hh[gi], d1[gi], lop[gi]
    c0 = fma(R, yr, H * xr)                              # w at the tile's origin (staging lane)
    regular = D >= 0
    # shapes: (P, 16 x, 4 anchors, 4 j)
    cx = np.arange(16, dtype=f32)[None, :, None, None]
    cy0 = (4 * np.arange(4, dtype=f32))[None, None, :, None]
    j = np.arange(4, dtype=f32)[None, None, None, :]
    e = lambda a: a[:, None, None, None]  # noqa: E731
    # float64 from the float32 record
    ex64 = e(xr).astype(f64) - cx
    w64 = e(R).astype(f64) * (e(yr).astype(f64) - (cy0 + j)) + e(H).astype(f64) * ex64
    E64 = e(L).astype(f64) - e(D).astype(f64) * ex64 * ex64 - w64 * w64
    a64 = np.exp2(E64)
    # direct float32 (round 3)
    e_x = (e(xr) - cx).astype(f32)
    s0 = fma(-(e(D) * e_x).astype(f32), e_x, e(L) + 0 * e_x)
    c = fma(-e(H) + 0 * cx, cx + 0 * e(H), e(c0) + 0 * cx)
    wd = fma(-e(R) + 0 * c + 0 * j, (cy0 + j) + 0 * c, c + 0 * j)
    a_dir = exp2_32(fma(-wd, wd, s0 + 0 * wd))
    # recurrence float32
    w0 = fma(-e(R) + 0 * c, cy0 + 0 * c, c + 0 * cy0)
    a0 = exp2_32(fma(-w0, w0, s0 + 0 * w0))
    tr = (e(R) + e(R)).astype(f32)
    nr2 = (-(e(R) * e(R))).astype(f32)
    hx = exp2_32((nr2 + nr2).astype(f32))
    G0 = np.minimum(fma(tr + 0 * w0, w0, nr2 + 0 * w0), f32(args.gmax))
    g0 = exp2_32(G0)
    if args.products:   # pixel-pair products: (a2, a3) = (a0, a1) * (g0^2 hx, g0^2 hx^3)
        hx3 = exp2_32((f32(6.0) * nr2).astype(f32))
        a1 = (a0 * g0).astype(f32)
        gg = (g0 * g0).astype(f32)
        a2 = (a0 * (gg * hx).astype(f32)).astype(f32)
        a3 = (a1 * (gg * hx3).astype(f32)).astype(f32)
    else:
        a1 = (a0 * g0).astype(f32)
        g1 = (g0 * hx).astype(f32)
        a2 = (a1 * g1).astype(f32)
        g2 = (g1 * hx).astype(f32)
        a3 = (a2 * g2).astype(f32)
    a_rec = np.concatenate([a0, a1, a2, a3], axis=3)
    bad = ~np.isfinite(a_rec)
    print("non-finite recurrence alphas:", int(bad.sum()))
    # what the kernel stages: bound over the tile >= -26 (approximate with the exact max over the tile)
    staged = (E64.reshape(P, -1).max(axis=1) >= -26) & regular
    print("staged pairs %d (%.1f %%), regular %.1f %%" % (staged.sum(), 100 * staged.mean(), 100 * regular.mean()))
    err_d = np.abs(a_dir - a64).reshape(P, -1).max(axis=1)
    err_r = np.abs(a_rec - a64).reshape(P, -1).max(axis=1)
    G0max = np.abs(fma(tr + 0 * w0, w0, nr2 + 0 * w0)).reshape(P, -1).max(axis=1)
    print("max |alpha error| over staged pairs: direct %.3g   recurrence %.3g" % (err_d[staged].max(), err_r[staged].max()))
    edges = [0, 0.5, 0.75, 1.0, 1.25, 1.5, 1.75, 2.0, 2.5, 3.0, 4.0, 1e9]
    print("%-14s %9s %8s %12s %12s %12s %10s" % ("r11", "pairs", "share", "direct max", "recur max", "recur p99.9", "max|G0|"))
    for lo, hi in zip(edges[:-1], edges[1:]):
        m = staged & (R >= lo) & (R < hi)
        if not m.any():
            continue
        print("[%4.2f,%5.2f) %9d %7.2f%% %12.3g %12.3g %12.3g %10.1f" % (
            lo, min(hi, 99), m.sum(), 100 * m.sum() / staged.sum(), err_d[m].max(), err_r[m].max(),
            np.quantile(err_r[m], 0.999), G0max[m].max()))
    # the same by the largest |G0| over the tile (a per-(record, tile) rule the staging lane could apply)
    print("by max |G0| over the tile:")
    for lo, hi in zip([0, 8, 16, 24, 32, 48, 64, 128], [8, 16, 24, 32, 48, 64, 128, 1e9]):
        m = staged & (G0max >= lo) & (G0max < hi)
        if not m.any():
            continue
        print("[%5.0f,%5.0f) %9d %7.2f%% direct %10.3g recur %10.3g" % (lo, min(hi, 99999), m.sum(), 100 * m.sum() / staged.sum(),
                                                                       err_d[m].max(), err_r[m].max()))
    # summed relative effect: error weighted like a pixel sees it (sum over a tile's records is what matters)
    print("mean |alpha error| staged: direct %.3g recurrence %.3g" % (
        np.abs(a_dir - a64)[staged].mean(), np.abs(a_rec - a64)[staged].mean()))


if __name__ == "__main__":
    main()
