"""What does the compositing launch (blend_tile16_ref_kernel) spend its time on?  One frame of a bench workload on
libgsx_test.so with the blend probe on: per tile its batches walked, the first batch that held a reference-order
record, how many batches / records did.   python tools/attic/ref_probe.py [workload]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from intro_to_gaussian_splatting_amd import _ffi
_ffi.use_test_library()
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c3_clustered"
sc, scene = bench.build_scene(wl, "cuda")
lib = _ffi.load()
for _ in range(3):
    scene.render_image_hip(1)
grid = 1 << 17                # gsx_blend.hip: kProbeSecond
buf = torch.zeros((4 * grid, 4), dtype=torch.int32, device="cuda")
lib.gsx_debug_set_blend_probe(buf.data_ptr())
st = {}
scene.render_image_hip(1, stats=st, timing=True)
torch.cuda.synchronize()
lib.gsx_debug_set_blend_probe(None)
full = buf.cpu().numpy().view(np.uint32)
d = full[2 * grid:2 * grid + 65536]
cyc4 = full[2 * grid + 65536:3 * grid].astype(np.int64)       # staging, compositing regular / wild / ref-order batches
ent3 = full[3 * grid:3 * grid + 65536].astype(np.int64)
used = np.nonzero(d[:, 3])[0]
cyc = d[used, 0].astype(np.int64)
batches = d[used, 1] & 0xFFF
first = (d[used, 1] >> 12) & 0xFFF
sat = d[used, 1] >> 31
refb = d[used, 2] & 0xFFF
refr = d[used, 2] >> 12
ln = d[used, 3]
print(wl, "blend stage %.3f ms, D %d, tiles redone %d (n_redo %s)" % (st["stage_ms"]["blend"], st["n_instances"], len(used), st.get("n_redo")))
q = lambda a: np.percentile(a, [10, 50, 90, 99, 100]).round(1).tolist()
print("list length            p10/50/90/99/max", q(ln))
print("batches walked                         ", q(batches), "sum", int(batches.sum()))
print("first ref-order batch                  ", q(first[first < 0xFFF]), "tiles with none:", int((first == 0xFFF).sum()))
print("first / walked                         ", q(first[first < 0xFFF] / np.maximum(batches[first < 0xFFF], 1)))
print("ref-order batches per tile             ", q(refb), "sum", int(refb.sum()), "= %.1f %% of batches walked" % (100.0 * refb.sum() / max(batches.sum(), 1)))
print("ref-order records per ref-order batch  ", q(refr[refb > 0] / refb[refb > 0]), "sum", int(refr.sum()))
print("cycles per tile (us @2.4 GHz)          ", q(cyc / 2400.0), "sum %.1f ms-waves" % (cyc.sum() / 2.4e6))
print("saturated tiles %d" % int(sat.sum()))
c = cyc4[used].sum(axis=0)
e = ent3[used].sum(axis=0)
print("cycles: staging %.1f %%, compositing regular %.1f %% / wild %.1f %% / ref-order %.1f %% of %.1f ms-waves accounted" % (
    100.0 * c[0] / c.sum(), 100.0 * c[1] / c.sum(), 100.0 * c[2] / c.sum(), 100.0 * c[3] / c.sum(), c.sum() / 2.4e6))
print("entries walked: regular %d, wild %d, ref-order %d;  cycles per entry: %.0f / %.0f / %.0f" % (
    e[0], e[1], e[2], c[1] / max(e[0], 1), c[2] / max(e[1], 1), c[3] / max(e[2], 1)))
# long tiles' helper workgroups (in-place kernel, one wave each): first record by block, cycle split at 3 grid + 65536 + block
h = full[:grid]
hq = full[3 * grid + 65536:4 * grid].astype(np.int64)
helpers = np.nonzero(h[:65536, 1] & 0x40000000)[0]
if len(helpers):
    hc = h[helpers, 0].astype(np.int64)
    print("long tiles' quarters: %d, cycles (us) p50/90/max %s, list length p50/max %s, staged p50 %d" % (
        len(helpers), np.percentile(hc / 2400.0, [50, 90, 100]).round(1).tolist(), np.percentile(h[helpers, 2], [50, 100]).tolist(),
        np.median(h[helpers, 3] & 0x7FFFFFFF)))
    qs = hq[helpers].sum(axis=0)
    nref, ntr = (qs[3] & 0xFFF), 0
    refs = (hq[helpers, 3] & 0xFFF).sum(); trips = (hq[helpers, 3] >> 12).sum()
    tot = qs[:3].sum()
    print("  cycles: staging %.1f %%, plain trips %.1f %% (%d trips, %.0f cycles each), ref-order trips %.1f %% (%d, %.0f each)" % (
        100.0 * qs[0] / tot, 100.0 * qs[1] / tot, trips, qs[1] / max(trips, 1), 100.0 * qs[2] / tot, refs, qs[2] / max(refs, 1)))
    worst = helpers[np.argsort(-hc)[:6]]
    for i in worst:
        print("   block %5d tile %5d: %.1f us, list %d, staged %d; staging %.1f us, plain trips %.1f us, ref trips %.1f us (%d / %d)" % (
            i, h[i, 1] & 0xFFFFFF, h[i, 0] / 2400.0, h[i, 2], h[i, 3] & 0x7FFFFFFF, hq[i, 0] / 2400.0, hq[i, 1] / 2400.0, hq[i, 2] / 2400.0,
            hq[i, 3] & 0xFFF, hq[i, 3] >> 12))
