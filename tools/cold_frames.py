"""Frames WITHOUT hints (what a view's first frame runs: sample, whole-frame schedule, separate launches), for a kernel trace:
    rocprofv3 --kernel-trace --stats --output-format csv -d DIR -o c -- python3 tools/cold_frames.py [workload] [frames]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 200
sc, scene = bench.build_scene(wl, "cuda")
out = None
for k in range(frames + 5):
    out = scene.render_image_hip(1, out=out, no_sync=True, use_hints=False)
    if k % 50 == 0:
        torch.cuda.synchronize()
        scene.confirm_frames()
torch.cuda.synchronize()
scene.confirm_frames()
