"""What ends the compositing launch of a heavy-tailed view?  One warm frame of a bench workload on libgsx_test.so with the
blend probe on (csrc/gsx_debug.h; REF instance: per-tile records at 2 * kProbeSecond + tile, per-helper records at the
helper's block index): the slowest single-wave tiles against the slowest quarters of long tiles, in shader cycles.
    python tools/attic/long_tail.py [workload]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from intro_to_gaussian_splatting_amd import _ffi
_ffi.use_test_library()
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c3_clustered"
sc, scene = bench.build_scene(wl, "cuda")
lib = _ffi.load()
for _ in range(8):
    scene.render_image_hip(1)
half = 1 << 17
buf = torch.zeros((4 * half, 4), dtype=torch.int32, device="cuda")
lib.gsx_debug_set_blend_probe(buf.data_ptr())
st = {}
scene.render_image_hip(1, stats=st, timing=True)
torch.cuda.synchronize()
lib.gsx_debug_set_blend_probe(None)
d = buf.cpu().numpy().view(np.uint32)
print(wl, "blend stage %.3f ms, D %d" % (st["stage_ms"]["blend"], st["n_instances"]))
tiles = d[2 * half:2 * half + 65536]
t_ids = np.nonzero(tiles[:, 3])[0]
cyc = tiles[t_ids, 0].astype(np.int64)
order = np.argsort(-cyc)
print("single-wave tiles: %d; cycles p50 %d p99 %d max %d" % (t_ids.size, np.percentile(cyc, 50), np.percentile(cyc, 99), cyc.max()))
print("slowest single-wave tiles: tile cycles batches list-length")
for i in order[:16]:
    pc = d[2 * half + 65536 + t_ids[i]]
    pe = d[3 * half + t_ids[i]]
    print("   tile %5d  %8d cycles  %4d batches  list %6d   cycles staging %7d, regular trips %7d, mono %6d, ref-order %7d; entries walked %5d / %5d / %5d" % (
        t_ids[i], cyc[i], tiles[t_ids[i], 1] & 0xFFF, tiles[t_ids[i], 3], pc[0], pc[1], pc[2], pc[3], pe[0], pe[1], pe[2]))
q = d[:half]
qi = np.nonzero(q[:, 1] & 0x40000000)[0]
qc = q[qi, 0].astype(np.int64)
print("quarters of long tiles: %d (= %d tiles); cycles p50 %d max %d" % (qi.size, qi.size // 4, np.percentile(qc, 50) if qi.size else 0, qc.max() if qi.size else 0))
sec = d[half + qi]
if qi.size:
    start, end = sec[:, 3].astype(np.int64), sec[:, 1].astype(np.int64)
    t0 = start.min()
    print("helpers' wall clock: first start 0, last start %.1f us, last end %.1f us" % ((start.max() - t0) * 0.01, (end.max() - t0) * 0.01))
    o = np.argsort(-qc)
    print("slowest quarters: tile quarter-block cycles list staged start-us end-us")
    for i in o[:16]:
        pq = d[3 * half + 65536 + qi[i]]
        print("   tile %5d  block %6d  %8d cycles  list %6d  staged %6d  %.1f .. %.1f us   cycles staging %7d, plain trips %7d, ref trips %7d; trips %d, ref records %d" % (
            q[qi[i], 1] & 0x3FFFFFFF, qi[i], qc[i], q[qi[i], 2], q[qi[i], 3] & 0x7FFFFFFF, (start[i] - t0) * 0.01, (end[i] - t0) * 0.01,
            pq[0], pq[1], pq[2], pq[3] >> 12, pq[3] & 0xFFF))
