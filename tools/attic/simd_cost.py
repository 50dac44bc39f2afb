"""Does a SIMD's finishing time follow the cost of the waves it ran?  One converged frame of a bench workload on
libgsx_test.so with the blend probe on; tiles AND helper workgroups, cost = entries walked + 5 per batch.
   python tools/attic/simd_cost.py [workload]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from intro_to_gaussian_splatting_amd import _ffi
_ffi.use_test_library()
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c3_clustered"
sc, scene = bench.build_scene(wl, "cuda")
lib = _ffi.load()
for _ in range(12):
    scene.render_image_hip(1)
torch.cuda.synchronize()
half = 1 << 17
buf = torch.zeros((4 * half, 4), dtype=torch.int32, device="cuda")
lib.gsx_debug_set_blend_probe(buf.data_ptr())
st = {}
scene.render_image_hip(1, stats=st, timing=True)
torch.cuda.synchronize()
lib.gsx_debug_set_blend_probe(None)
d = buf.cpu().numpy().view(np.uint32)
rows = np.nonzero(d[:half, 2] > 0)[0]
sec = d[half + rows]
helper = (d[rows, 1] & 0x40000000) != 0
ln = d[rows, 2].astype(np.int64)
walked = (d[rows, 3] & 0x7FFFFFFF).astype(np.int64)
cost = walked + 5 * ((ln + 63) // 64)
start, end = sec[:, 3].astype(np.int64), sec[:, 1].astype(np.int64)
t0 = start.min()
start, end = (start - t0) * 0.01, (end - t0) * 0.01
simd = ((sec[:, 2] >> 16).astype(np.int64) << 16) | (sec[:, 2] & 0xFF30)
keys, inv = np.unique(simd, return_inverse=True)
load = np.bincount(inv, weights=cost)
nw = np.bincount(inv)
nh = np.bincount(inv, weights=helper.astype(float))
last = np.zeros(len(keys)); np.maximum.at(last, inv, end)
longest = np.zeros(len(keys)); np.maximum.at(longest, inv, end - start)
batches = np.zeros(len(keys)); np.maximum.at(batches, inv, (ln + 63) // 64)
print(wl, "blend %.3f ms; %d SIMDs, %d tile + %d helper workgroups; span %.1f us" % (st["stage_ms"]["blend"], len(keys), (~helper).sum(), helper.sum(), end.max()))
print("cost per SIMD: mean %.0f max/mean %.3f min/mean %.3f;  corr(last end, cost) %.2f;  corr(last end, most batches of one wave) %.2f" % (
    load.mean(), load.max() / load.mean(), load.min() / load.mean(), np.corrcoef(last, load)[0, 1], np.corrcoef(last, batches)[0, 1]))
o = np.argsort(-last)[:10]
print("SIMDs that end last: end us | cost / mean | waves (helpers) | longest wave us | most batches in one wave")
for i in o:
    print("   %6.1f | %.2f | %d (%d) | %6.1f | %d" % (last[i], load[i] / load.mean(), nw[i], nh[i], longest[i], batches[i]))
q = np.argsort(load)
print("by cost decile: mean end us " + " ".join("%.0f" % last[q[k * len(q) // 10:(k + 1) * len(q) // 10]].mean() for k in range(10)))
x0 = (keys >> 16) == 0
print("helpers per SIMD on XCD 0 (SIMD order of first appearance): " + "".join(str(int(v)) for v in nh[x0]))
# placement: helper workgroup j and tile workgroup i of an XCD's share -- which SIMD?  (grid: [rank groups unless last]
# [helpers] [tiles] ..; XCD = block & 7, index inside the share = block >> 3)
hb = rows[helper]; tb = rows[~helper]
h0 = hb.min() & ~7 if helper.any() else 0
t0b = tb.min() & ~7
print("first helper block %d, first tile block %d (helper positions per XCD: %d)" % (h0, t0b, (t0b - h0) >> 3))
for x in (0, 3):
    hs = {int((b - h0) >> 3): int(s) for b, s in zip(hb, simd[helper]) if (b & 7) == x}
    ts = {int((b - t0b) >> 3): int(s) for b, s in zip(tb, simd[~helper]) if (b & 7) == x}
    best = max(range(128), key=lambda k: sum(1 for j, s in hs.items() if ts.get((j + k) % 128) == s or ts.get((j + k) % 128 + 128) == s))
    for k in (0, best):
        m = sum(1 for j, s in hs.items() if any(ts.get(((j + k) % 128) + 128 * r) == s for r in range(8)))
        print("  XCD %d: helpers %d; share whose SIMD also runs tile workgroups i == j + %d (mod 128): %.2f" % (x, len(hs), k, m / max(len(hs), 1)))
    per = sum(1 for i, s in ts.items() if i + 128 in ts and ts[i + 128] == s) / max(1, sum(1 for i in ts if i + 128 in ts))
    print("  XCD %d: tiles i and i + 128 on the same SIMD: %.2f; tile entries %d" % (x, per, len(ts)))
# does the hand-out arrive?  cost per residue (i % 128) of the tile workgroups' index in their XCD's share, and how pure
# the residue -> SIMD mapping is
for x in (0, 5):
    mt = (~helper) & ((rows & 7) == x)
    i = (rows[mt] - t0b) >> 3
    res = i % 128
    lr = np.bincount(res, weights=cost[mt], minlength=128)
    sim = simd[mt]
    pure = []
    for r in range(128):
        ss = sim[res == r]
        if len(ss):
            pure.append(np.bincount(np.unique(ss, return_inverse=True)[1]).max() / len(ss))
    # and by round: share of workgroups of round k that sit on the SIMD of their residue's round-0 workgroup
    base = {int(r): int(s) for r, s, k in zip(res, sim, i >> 7) if k == 0}
    by_round = [np.mean([base.get(int(r)) == int(s) for r, s, k in zip(res, sim, i >> 7) if k == rr]) for rr in range(int(i.max() >> 7) + 1)]
    print("  XCD %d: tile cost per residue max/mean %.3f min/mean %.3f; a residue's workgroups on one SIMD: mean share %.2f; by round %s" % (
        x, lr.max() / lr.mean(), lr.min() / lr.mean(), np.mean(pure), " ".join("%.2f" % v for v in by_round)))
