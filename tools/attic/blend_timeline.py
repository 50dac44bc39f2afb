"""When does every workgroup of the compositing launch run?  One frame on libgsx_test.so with the blend probe on; prints
the number of workgroups in flight per 20 us of the launch and the last ones to finish.
    python tools/attic/blend_timeline.py [workload]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from intro_to_gaussian_splatting_amd import _ffi
_ffi.use_test_library()
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c3_clustered"
sc, scene = bench.build_scene(wl, "cuda")
lib = _ffi.load()
for _ in range(int(os.environ.get("WARM", "40"))):
    scene.render_image_hip(1)
grid = 1 << 17
buf = torch.zeros((4 * grid, 4), dtype=torch.int32, device="cuda")
lib.gsx_debug_set_blend_probe(buf.data_ptr())
st = {}
scene.render_image_hip(1, stats=st, timing=True)
torch.cuda.synchronize()
lib.gsx_debug_set_blend_probe(None)
full = buf.cpu().numpy().view(np.uint32)
a, b = full[:grid], full[grid:2 * grid]
used = np.nonzero(a[:, 2])[0]
t1 = b[used, 1].astype(np.int64) * 10          # ns
t0 = b[used, 3].astype(np.int64) * 10
helper = (a[used, 1] & 0x40000000) != 0
z = t0.min()
t0, t1 = t0 - z, t1 - z
print(wl, "blend stage %.3f ms; %d workgroups probed (%d helpers); first start 0, last end %.1f us" % (
    st["stage_ms"]["blend"], len(used), int(helper.sum()), t1.max() / 1000.0))
edges = np.arange(0, t1.max() + 20000, 20000)
for lo in edges[:-1]:
    hi = lo + 20000
    act = ((t0 < hi) & (t1 > lo))
    print("  %4d..%4d us: %5d in flight (%4d helpers), %5d started, %5d ended" % (
        lo // 1000, hi // 1000, int(act.sum()), int((act & helper).sum()), int(((t0 >= lo) & (t0 < hi)).sum()), int(((t1 >= lo) & (t1 < hi)).sum())))
last = np.argsort(-t1)[:10]
for i in last:
    print("   block %5d tile %5d %s: %.1f .. %.1f us, list %d, staged %d" % (
        used[i], a[used[i], 1] & 0xFFFFFF, "Q" if helper[i] else " ", t0[i] / 1000.0, t1[i] / 1000.0, a[used[i], 2], a[used[i], 3] & 0x7FFFFFFF))
tiles = used[~helper]
order = np.argsort(tiles)
print("tile workgroups in block order: block, tile, list, staged, start us")
for j in list(order[:12]) + list(order[len(order) // 2:len(order) // 2 + 6]) + list(order[-6:]):
    i = np.nonzero(~helper)[0][j]
    print("   %5d %5d %6d %6d %7.1f" % (used[i], a[used[i], 1] & 0xFFFFFF, a[used[i], 2], a[used[i], 3] & 0x7FFFFFFF, t0[i] / 1000.0))
hb = hint_bytes = None
