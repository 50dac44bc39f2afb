"""Soak of the hint feedback (GsxParams.hints: splitters, tile costs, schedule, long-tile threshold -- every frame leaves them for
the next): tens of thousands of replays of one captured frame, camera at rest and jumping along / across an orbit, every frame
checked against the frame the same pose gave the first time.  Hints may cost time, never a pixel.
    python tools/soak.py [workload] [replays at rest] [orbit passes]"""
import os
import sys
import time
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
n_rest = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 40
dev = "cuda:0"
sc, scene = bench.build_scene(wl, dev, orbit=30)
ids = sorted(i for i in scene.images if i != 1)
mid = ids[len(ids) // 2]
frame = scene.capture_frame(mid, movable_camera=True, headroom=1.4)


def digest(t: torch.Tensor):
    # (a 64-bit sum of the frame's words on the device: one scalar crosses PCIe per check)
    return int(t.view(torch.int32).to(torch.int64).sum().item())


want = {}
for i in ids:                               # every pose once, from whatever the previous pose left: the frames to reproduce
    frame.set_camera(i)
    frame.replay()
    want[i] = (digest(frame.confirm()), frame.counts()[:2])
    alone = scene.render_image_hip(i, use_hints=False)
    assert torch.equal(alone, frame.out), ("a captured frame differs from the frame rendered from scratch", i)
t0 = time.time()
frame.set_camera(mid)
bad = 0
for k in range(n_rest):                     # at rest: the feedback loop runs on its own output
    frame.replay()
    if k % 500 == 499:
        bad += digest(frame.confirm()) != want[mid][0]
print("%s: %d replays at rest in %.1f s, %d checked, %d differing" % (wl, n_rest, time.time() - t0, n_rest // 500, bad), flush=True)
rs = np.random.RandomState(0)
t0 = time.time()
checked = 0
for p in range(passes):
    seq = ids + ids[-2:0:-1] if p % 2 == 0 else [int(v) for v in rs.choice(ids, size=2 * len(ids))]      # one step at a time / jumps
    for i in seq:
        frame.set_camera(i)
        frame.replay()
        if rs.uniform() < 0.2:
            d = digest(frame.confirm())
            checked += 1
            bad += d != want[i][0] or frame.counts()[:2] != want[i][1]
print("%s: %d orbit passes (%d frames, every other pass in random jumps) in %.1f s, %d checked, %d differing" % (
    wl, passes, passes * 2 * len(ids), time.time() - t0, checked, bad), flush=True)
sys.exit(1 if bad else 0)
