"""How does the compositing kernel's time scale with the number of resident waves per SIMD?
Renders tile windows of C3 holding ~1, 2, 4, 6, 8 single-wave workgroups per SIMD (1024 SIMDs) and
prints the compositing stage time (HIP events, GSX_FLAG_TIMING).  Throughput-bound => time grows
linearly with the tile count; latency-bound => flat."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_scene

sc, scene = build_scene("c3", "cuda:0")
nty = 67
for cols in (15, 30, 61, 91, 119):
    best = 1e9
    for _ in range(5):
        st = {}
        scene.render_image_hip(1, tile_window=(0, cols, 0, nty), stats=st, timing=True)
        best = min(best, st["stage_ms"]["blend"])
    tiles = cols * nty
    print("tiles %5d  (%.2f waves/SIMD)  D %8d  blend %.4f ms  -> %.1f ns per instance" %
          (tiles, tiles / 1024.0, st["n_instances"], best, best * 1e6 / st["n_instances"]))
