"""What ends the compositing launch?  One converged frame (12 warm-up frames: hints settled) of a bench workload on
libgsx_test.so with the blend probe on: the workgroups that END last -- wall-clock start / end, tile, helper?, list
length, entries walked, saturated?.   python tools/attic/blend_tail.py [workload]"""
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
start, end = sec[:, 3].astype(np.int64), sec[:, 1].astype(np.int64)
ok = helper | (end > 0)
t0 = start[~helper].min()
print(wl, "blend %.3f ms; %d tile workgroups, %d helper workgroups" % (st["stage_ms"]["blend"], (~helper).sum(), helper.sum()))
m = ~helper          # (helpers leave no second record)
s_us, e_us = (start[m] - t0) * 0.01, (end[m] - t0) * 0.01
r = rows[m]
order = np.argsort(-e_us)[:16]
print("last to end (tile workgroups): end us, start us, tile, list, walked, saturated, batches")
for i in order:
    print("  %7.1f %7.1f  tile %5d  len %6d  walked %5d  sat %d  batches %d" % (
        e_us[i], s_us[i], d[r[i], 1] & 0x3FFFFFFF, d[r[i], 2], d[r[i], 3] & 0x7FFFFFFF, d[r[i], 3] >> 31, -(-int(d[r[i], 2]) // 64)))
print("span %.1f us; p50 end %.1f, p90 %.1f, p99 %.1f" % (e_us.max(), np.percentile(e_us, 50), np.percentile(e_us, 90), np.percentile(e_us, 99)))
cyc = d[rows[helper], 0].astype(np.float64)
if helper.any():
    hr = rows[helper]
    o = np.argsort(-cyc)[:8]
    print("helpers by cycles: " + ", ".join("tile %d len %d walked %d %.0f us@2GHz" % (d[hr[i], 1] & 0x3FFFFFFF, d[hr[i], 2], d[hr[i], 3] & 0x7FFFFFFF, cyc[i] / 2000.0) for i in o))
