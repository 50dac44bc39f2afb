"""How many (Gaussian, tile) pairs of a bench scene contribute nothing anywhere in their tile -- at the
16x16 tile of the binning and at 8x8 quadrants of it?  CPU only (oracle projection + numpy):
    python tools/attic/analyze_skip.py [workload]
A pair is negligible when ln(alpha) = ln(opacity) + max over the pixel block of the exponent < -26 ln 2
(the staging test of blend_tile16_kernel)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from intro_to_gaussian_splatting_amd.synthetic import make_scene  # noqa: E402
from oracle import c_oracle, cpu_ref  # noqa: E402


def min_quadratic_over_box(a, b, c, mx, my, x0, x1, y0, y1):
    """min over [x0,x1]x[y0,y1] of a ex^2 + 2 b ex ey + c ey^2, (ex, ey) = (x - mx, y - my); PSD."""
    cx = np.clip(mx, x0, x1)
    cy = np.clip(my, y0, y1)
    inside = (cx == mx) & (cy == my)
    best = np.full(a.shape, np.inf)
    for fixed_x in (x0, x1):      # vertical edges: x fixed, minimise over y
        ex = fixed_x - mx
        ey = np.clip(-b * ex / np.maximum(c, 1e-30), y0 - my, y1 - my)
        best = np.minimum(best, a * ex * ex + 2 * b * ex * ey + c * ey * ey)
    for fixed_y in (y0, y1):
        ey = fixed_y - my
        ex = np.clip(-b * ey / np.maximum(a, 1e-30), x0 - mx, x1 - mx)
        best = np.minimum(best, a * ex * ex + 2 * b * ex * ey + c * ey * ey)
    return np.where(inside, 0.0, best)


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
    n, w, h, _ = bench.WORKLOADS[wl]
    sc = make_scene(n, w, h, seed=0, **bench.GENERATOR_ARGS.get(wl, {}))
    cam = cpu_ref.build_camera(sc["qvec"], sc["tvec"], sc["fx"], sc["fy"], w, h)
    pre = c_oracle.preprocess(sc["points"], sc["colors_0_255"] / 255.0, sc["scales"], sc["quaternions"], sc["opacity"], cam)
    tile = 16
    ntx, nty = len(cpu_ref.tile_origins(w, tile)), len(cpu_ref.tile_origins(h, tile))
    inv = pre.inverse_covariance_2d.astype(np.float64)
    a, b, c = inv[:, 0, 0], 0.5 * (inv[:, 0, 1] + inv[:, 1, 0]), inv[:, 1, 1]
    mx, my = pre.points_xy[:, 0].astype(np.float64), pre.points_xy[:, 1].astype(np.float64)
    ln_op = -np.log1p(np.exp(-pre.sigmoid_opacity[:, 0].astype(np.float64)))
    # tile rectangle of every Gaussian: x0 = 16 i with min_x <= x0 + 16 and max_x >= x0
    tx0 = np.maximum(np.ceil((pre.min_x.astype(np.float64) - tile) / tile), 0).astype(np.int64)
    tx1 = np.minimum(np.floor(pre.max_x.astype(np.float64) / tile), ntx - 1).astype(np.int64)
    ty0 = np.maximum(np.ceil((pre.min_y.astype(np.float64) - tile) / tile), 0).astype(np.int64)
    ty1 = np.minimum(np.floor(pre.max_y.astype(np.float64) / tile), nty - 1).astype(np.int64)
    nx, ny = np.maximum(tx1 - tx0 + 1, 0), np.maximum(ty1 - ty0 + 1, 0)
    cnt = nx * ny
    D = int(cnt.sum())
    g = np.repeat(np.arange(len(cnt)), cnt)
    k = np.arange(D) - np.repeat(np.cumsum(cnt) - cnt, cnt)
    px = (tx0[g] + k % np.maximum(nx[g], 1)) * tile
    py = (ty0[g] + k // np.maximum(nx[g], 1)) * tile
    thr = -26.0 * np.log(2.0)

    def negligible(x0, y0, side, side_y=None):
        side_y = side if side_y is None else side_y
        q = min_quadratic_over_box(a[g], b[g], c[g], mx[g], my[g], x0, x0 + side - 1, y0, y0 + side_y - 1)
        return ln_op[g] - 0.5 * q < thr

    neg16 = negligible(px, py, 16)
    print("%s: %d Gaussians kept, %d pairs (%.2f per Gaussian)" % (wl, len(cnt), D, D / max(len(cnt), 1)))
    print("negligible in the whole 16x16 tile: %.1f %% of pairs" % (100.0 * neg16.mean()))
    for side in (8, 4):
        live = 0
        m = 16 // side
        for qy in range(m):
            for qx in range(m):
                live += int((~negligible(px + side * qx, py + side * qy, side)).sum())
        print("%dx%d blocks: %.1f %% of the pixel work of the pairs is in non-negligible blocks" %
              (side, side, 100.0 * live / (D * m * m)))
    for bw, bh in ((16, 8), (8, 16), (16, 4), (16, 2), (16, 1)):
        live = 0
        for qy in range(16 // bh):
            for qx in range(16 // bw):
                live += int((~negligible(px + bw * qx, py + bh * qy, bw, bh)).sum())
        print("%dx%d blocks (w x h): %.1f %% of the pixel work is in non-negligible blocks" %
              (bw, bh, 100.0 * live / (D * (256 // (bw * bh)))))
    # the same with a 1/255-style cut, for orientation
    thr = np.log(1.0 / 255.0)
    print("(pairs whose alpha stays below 1/255 in the whole tile: %.1f %%)" % (100.0 * negligible(px, py, 16).mean()))


if __name__ == "__main__" and not (len(sys.argv) > 2 and sys.argv[2] in ("blocks", "bound")):
    main()


def block_lists(wl="c2"):
    """Round 4: if every 16-lane group of the wave (an 8x8 block of the tile, 4 pixels per lane) walked its OWN list of
    non-negligible records, a tile would cost max over its four blocks of the block's list length instead of the tile's
    list length.  Prints the mean of that ratio, weighted by list length."""
    n, w, h, _ = bench.WORKLOADS[wl]
    sc = make_scene(n, w, h, seed=0, **bench.GENERATOR_ARGS.get(wl, {}))
    cam = cpu_ref.build_camera(sc["qvec"], sc["tvec"], sc["fx"], sc["fy"], w, h)
    pre = c_oracle.preprocess(sc["points"], sc["colors_0_255"] / 255.0, sc["scales"], sc["quaternions"], sc["opacity"], cam)
    tile = 16
    ntx, nty = len(cpu_ref.tile_origins(w, tile)), len(cpu_ref.tile_origins(h, tile))
    inv = pre.inverse_covariance_2d.astype(np.float64)
    a, b, c = inv[:, 0, 0], 0.5 * (inv[:, 0, 1] + inv[:, 1, 0]), inv[:, 1, 1]
    mx, my = pre.points_xy[:, 0].astype(np.float64), pre.points_xy[:, 1].astype(np.float64)
    ln_op = -np.log1p(np.exp(-pre.sigmoid_opacity[:, 0].astype(np.float64)))
    tx0 = np.maximum(np.ceil((pre.min_x.astype(np.float64) - tile) / tile), 0).astype(np.int64)
    tx1 = np.minimum(np.floor(pre.max_x.astype(np.float64) / tile), ntx - 1).astype(np.int64)
    ty0 = np.maximum(np.ceil((pre.min_y.astype(np.float64) - tile) / tile), 0).astype(np.int64)
    ty1 = np.minimum(np.floor(pre.max_y.astype(np.float64) / tile), nty - 1).astype(np.int64)
    nx, ny = np.maximum(tx1 - tx0 + 1, 0), np.maximum(ty1 - ty0 + 1, 0)
    cnt = nx * ny
    D = int(cnt.sum())
    g = np.repeat(np.arange(len(cnt)), cnt)
    k = np.arange(D) - np.repeat(np.cumsum(cnt) - cnt, cnt)
    tx, ty = tx0[g] + k % np.maximum(nx[g], 1), ty0[g] + k // np.maximum(nx[g], 1)
    tid = tx * nty + ty
    thr = -26.0 * np.log(2.0)
    per_block = []
    for (bw, bh, label) in ((8, 8, "8x8 blocks (16 lanes x 4 px)"), (16, 4, "16x4 strips"), (4, 16, "4x16 strips"), (8, 4, "8x4 blocks"),
                            (4, 8, "4x8 blocks"), (4, 4, "4x4 blocks")):
        counts = []
        for qy in range(16 // bh):
            for qx in range(16 // bw):
                q = min_quadratic_over_box(a[g], b[g], c[g], mx[g], my[g], tx * 16 + bw * qx, tx * 16 + bw * qx + bw - 1,
                                           ty * 16 + bh * qy, ty * 16 + bh * qy + bh - 1)
                live = ln_op[g] - 0.5 * q >= thr
                counts.append(np.bincount(tid[live], minlength=ntx * nty))
        counts = np.stack(counts)
        full = np.bincount(tid, minlength=ntx * nty)
        print("%s, %s: sum over tiles of max block list / sum of tile lists = %.3f (mean block list %.3f)" % (
            wl, label, counts.max(axis=0).sum() / full.sum(), counts.mean(axis=0).sum() / full.sum()))


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "blocks":
    block_lists(sys.argv[1])


def kernel_bound(wl="c2"):
    """How much does the kernel's SEPARABLE bound (D1 min e0^2 + min w^2 over the block, stage_records) keep beyond the exact
    minimum of the quadratic over the block?  Prints kept shares per 8x8 block for both."""
    n, w, h, _ = bench.WORKLOADS[wl]
    sc = make_scene(n, w, h, seed=0, **bench.GENERATOR_ARGS.get(wl, {}))
    cam = cpu_ref.build_camera(sc["qvec"], sc["tvec"], sc["fx"], sc["fy"], w, h)
    pre = c_oracle.preprocess(sc["points"], sc["colors_0_255"] / 255.0, sc["scales"], sc["quaternions"], sc["opacity"], cam)
    tile = 16
    ntx, nty = len(cpu_ref.tile_origins(w, tile)), len(cpu_ref.tile_origins(h, tile))
    inv = pre.inverse_covariance_2d.astype(np.float64)
    a, b, c = inv[:, 0, 0], 0.5 * (inv[:, 0, 1] + inv[:, 1, 0]), inv[:, 1, 1]
    mx, my = pre.points_xy[:, 0].astype(np.float64), pre.points_xy[:, 1].astype(np.float64)
    ln_op = -np.log1p(np.exp(-pre.sigmoid_opacity[:, 0].astype(np.float64)))
    tx0 = np.maximum(np.ceil((pre.min_x.astype(np.float64) - tile) / tile), 0).astype(np.int64)
    tx1 = np.minimum(np.floor(pre.max_x.astype(np.float64) / tile), ntx - 1).astype(np.int64)
    ty0 = np.maximum(np.ceil((pre.min_y.astype(np.float64) - tile) / tile), 0).astype(np.int64)
    ty1 = np.minimum(np.floor(pre.max_y.astype(np.float64) / tile), nty - 1).astype(np.int64)
    nx, ny = np.maximum(tx1 - tx0 + 1, 0), np.maximum(ty1 - ty0 + 1, 0)
    cnt = nx * ny
    D = int(cnt.sum())
    g = np.repeat(np.arange(len(cnt)), cnt)
    k = np.arange(D) - np.repeat(np.cumsum(cnt) - cnt, cnt)
    tx, ty = tx0[g] + k % np.maximum(nx[g], 1), ty0[g] + k // np.maximum(nx[g], 1)
    thr = -26.0 * np.log(2.0)
    # completed square in y (gsx_project.hip pack_record): M = Q/2 (natural log units here), r11 = sqrt(M11), hh = M01 / r11, D1 = M00 - hh^2
    m00, m01, m11 = 0.5 * a[g], 0.5 * b[g], 0.5 * c[g]
    r11 = np.sqrt(m11)
    hh = m01 / r11
    d1 = m00 - hh * hh
    kept_exact = kept_sep = 0
    for qy in range(2):
        for qx in range(2):
            x0, y0 = tx * 16 + 8 * qx, ty * 16 + 8 * qy
            q = min_quadratic_over_box(a[g], b[g], c[g], mx[g], my[g], x0, x0 + 7, y0, y0 + 7)
            kept_exact += int((ln_op[g] - 0.5 * q >= thr).sum())
            ex0, ex1, ey0, ey1 = mx[g] - x0, mx[g] - (x0 + 7), my[g] - y0, my[g] - (y0 + 7)
            ex_min2 = np.where(ex0 * ex1 <= 0, 0.0, np.minimum(ex0 * ex0, ex1 * ex1))
            ws = np.stack([r11 * ey0 + hh * ex0, r11 * ey1 + hh * ex0, r11 * ey0 + hh * ex1, r11 * ey1 + hh * ex1])
            wlo, whi = ws.min(axis=0), ws.max(axis=0)
            w_min2 = np.where((wlo <= 0) & (whi >= 0), 0.0, np.minimum(np.abs(wlo), np.abs(whi)) ** 2)
            kept_sep += int((ln_op[g] - d1 * ex_min2 - w_min2 >= thr).sum())
    print("%s: (record, 8x8 block) combinations kept -- exact minimum of the quadratic: %.3f; the kernel's separable bound: %.3f" % (
        wl, kept_exact / (4 * D), kept_sep / (4 * D)))


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "bound":
    kernel_bound(sys.argv[1])
