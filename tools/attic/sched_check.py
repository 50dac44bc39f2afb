"""Is the schedule the projection launch's spare workgroups leave (GsxParams.hints) balanced by the costs it was made
from?  (Header word 5 = 1 marks the slot-cell layout of a round-4 experiment -- DESIGN.md section 5 --; the shipping layout
is every XCD's tiles by falling cost, dealt round by round by the compositing launch.)  Renders a bench workload until the hints have settled, reads the hints buffer back and adds up, per XCD
and SIMD slot, the costs (lens) of the tiles in its cells.   python tools/attic/sched_check.py [workload]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from intro_to_gaussian_splatting_amd import _ffi
_ffi.use_test_library()
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c3_clustered"
sc, scene = bench.build_scene(wl, "cuda")
for _ in range(12):
    scene.render_image_hip(1)
torch.cuda.synchronize()
(key, (buf, used)), = list(scene._hints.items())[-1:]
h = buf.cpu().numpy().view(np.uint32)
hdr = h[:64]
nt = int(hdr[2])
lens_off = 64 + 256 + 2048
lens = h[lens_off:lens_off + nt]
max_tiles = -(-int(sc['width']) // 16) * -(-int(sc['height']) // 16)      # (the buffer is laid out for the frame's tiles)
sched_off = lens_off + ((max_tiles * 4 + 255) & ~255) // 4
cut_k = max(1, nt // 256); s_ = nt // (8 * cut_k); cap = (s_ + 1) * cut_k; stride = (cap + 127) // 128 * 128 if hdr[5] == 1 else cap
print(wl, "tiles", nt, "kind", hdr[5], "sched for", hdr[3], "entries per XCD", hdr[8:16], "stride", stride)
long_ = (lens >> 31) != 0
cost = np.where(long_, 0, lens & 0x7FFFFFFF).astype(np.int64)
print("long tiles %d, mean helper cost %.0f; non-long cost: mean %.0f max %d" % (long_.sum(), (lens[long_] & 0x7FFFFFFF).mean() if long_.any() else 0, cost[~long_].mean(), cost.max()))
for x in range(8):
    n = int(hdr[8 + x])
    cells = h[sched_off + x * stride: sched_off + x * stride + n]
    if hdr[5] == 1:
        grid = cells.reshape(-1, 128)
        valid = grid < nt
        load = np.where(valid, cost[np.minimum(grid, nt - 1)], 0).sum(axis=0)
        cnt = valid.sum(axis=0)
        seen = np.sort(grid[valid])
    else:
        i = np.arange(n); r, s = i >> 7, i & 127
        in_round = np.minimum(128, n - (r << 7))
        k = (r << 7) + np.where(r & 1, in_round - 1 - s, s)
        load = np.bincount(s, weights=cost[cells[k]], minlength=128); cnt = np.bincount(s, minlength=128)
        seen = np.sort(cells)
    nlong = min(int(long_.sum()), 512); mine_long = (nlong - x + 7) // 8 if nlong > x else 0
    helpers = 4 * mine_long
    hl = (helpers // 128 + (np.arange(128) < helpers % 128)) * ((lens[long_] & 0x7FFFFFFF).mean() if long_.any() else 0)
    tot = load + hl
    print("  XCD %d: tiles %d (distinct %d); tile load max/mean %.3f; with helper loads: max/mean %.3f min/mean %.3f; tiles per slot %d..%d; on helper slots %.1f, others %.1f" % (
        x, len(seen), len(np.unique(seen)), load.max() / load.mean(), tot.max() / tot.mean(), tot.min() / tot.mean(), cnt.min(), cnt.max(),
        cnt[:helpers % 128 if helpers < 128 else 128].mean() if helpers else 0, cnt[helpers % 128:].mean() if helpers < 128 else 0))
if hdr[5] == 1:
    x = 0
    n = int(hdr[8 + x]); cells = h[sched_off: sched_off + n]; grid = cells.reshape(-1, 128)
    u, c = np.unique(cells[cells < nt], return_counts=True)
    dup = u[c > 1]
    print("XCD 0 duplicates:", len(dup), "first", dup[:10], "their cost", cost[dup[:10]], "long?", long_[dup[:10]])
    pos = [np.nonzero(cells == t)[0] for t in dup[:5]]
    print("positions (round, slot):", [[(int(p) >> 7, int(p) & 127) for p in q] for q in pos])
    nlong = min(int(long_.sum()), 512); helpers = 4 * ((nlong - x + 7) // 8)
    valid = grid < nt
    load = np.where(valid, cost[np.minimum(grid, nt - 1)], 0).sum(axis=0)
    print("helper slots:", helpers, "mean tile load on helper slots %.0f, on the others %.0f" % (load[:helpers % 128 if helpers < 128 else 128].mean(), load[helpers:].mean() if helpers < 128 else 0))
    print("first round of XCD 0 (costs):", cost[np.minimum(grid[0], nt - 1)][:32])
    print("last round of XCD 0 (costs):", np.where(grid[-1] < nt, cost[np.minimum(grid[-1], nt - 1)], -1)[:32])
