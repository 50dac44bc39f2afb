"""Which kernels pay for stale hints?  One movable-camera hipGraph of a bench workload (bench.py --camera-path), replayed
with the camera at rest (MODE=rest) or stepping a degree per frame along the orbit (MODE=moving) -- run each under
    rocprofv3 --kernel-trace --stats --output-format csv -d DIR -o o -- python3 tools/attic/moving_probe.py WORKLOAD MODE
and compare the two kernel tables (tools/kstats.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c3_clustered"
mode = sys.argv[2] if len(sys.argv) > 2 else "moving"
sc, scene = bench.build_scene(wl, "cuda", orbit=61)
ids = sorted(i for i in scene.images if i != 1)
mid = ids[len(ids) // 2]
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    frame = scene.capture_frame(mid, tile_size=16, movable_camera=True, headroom=1.3)
    seq = ids + ids[-2:0:-1]
    near = [i for i in seq if abs(i - mid) <= 6]          # stay near the middle pose: the same Gaussians in view
    for i in (near if mode == "moving" else [mid] * len(near)) * 12:
        frame.set_camera(i)
        frame.replay()
    torch.cuda.synchronize()
print(wl, mode, "done")
