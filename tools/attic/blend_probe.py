"""Who is the compositing launch waiting for?  Renders one frame of a bench workload on libgsx_test.so with the blend
probe on (csrc/gsx_debug.h) and prints the workgroups that ran longest: cycles, tile, list length, records staged,
whether / when the tile saturated.   python tools/attic/blend_probe.py [workload] [lib suffix]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from intro_to_gaussian_splatting_amd import _ffi
if len(sys.argv) > 2:
    os.environ["GSX_TEST_LIB_PATH"] = os.path.join(os.path.dirname(_ffi.LIB_PATH), "libgsx_test%s.so" % sys.argv[2])
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
d = buf.cpu().numpy().view(np.uint32)[:grid]       # (the second record of every workgroup lies behind: tools/attic/simd_balance.py)
cyc = d[:, 0].astype(np.int64)
used = np.nonzero(d[:, 2] | d[:, 1])[0]
print(wl, "blend stage %.3f ms, D %d" % (st["stage_ms"]["blend"], st["n_instances"]))
rows = [i for i in used if d[i, 2] > 0]
rows.sort(key=lambda i: -cyc[i])
print("slowest workgroups: block cycles(us@2.4GHz) tile long? length staged saturated")
for i in rows[:24]:
    print("%6d %9d (%6.1f us) tile %5d %s len %6d staged %6d sat %d" % (
        i, cyc[i], cyc[i] / 2400.0, d[i, 1] & 0x3FFFFFFF, "Q" if d[i, 1] & 0x40000000 else " ", d[i, 2], d[i, 3] & 0x7FFFFFFF, d[i, 3] >> 31))
tot = cyc[rows].sum()
print("sum of workgroup cycles %.3g; helpers %d, longest list %d" % (tot, sum(1 for i in rows if d[i, 1] & 0x40000000), d[rows, 2].max()))
