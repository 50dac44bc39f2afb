"""Why does a graph captured with a movable camera run slower at rest than one with the camera baked in?  Times four
captured frames of one view: camera baked in / movable x pair capacity 1.1 / 1.3 times the view's count.
    python tools/attic/movable_gap.py [workload]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c3_clustered"
sc, scene = bench.build_scene(wl, "cuda")
for _ in range(3):
    scene.render_image_hip(1)
for movable in (False, True):
    for headroom in (1.1, 1.3):
        frame = scene.capture_frame(1, movable_camera=movable, headroom=headroom)
        for _ in range(30):
            frame.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(200):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); frame.replay(); e1.record()
            ts.append((e0, e1))
        torch.cuda.synchronize()
        print("%s movable=%s headroom=%.1f: median %.4f ms (plain instance: %s)" % (
            wl, movable, headroom, float(np.median([a.elapsed_time(b) for a, b in ts])), frame._plain_footprints))
