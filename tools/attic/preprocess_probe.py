"""How long does GaussianScene.preprocess() take (the reference's stage-1 API, gsx_preprocess)?
    python tools/attic/preprocess_probe.py [workload]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
sc, scene = bench.build_scene(wl, "cuda")
for _ in range(5):
    pre = scene.preprocess(1)
torch.cuda.synchronize()
ts = []
for _ in range(30):
    t0 = time.perf_counter()
    pre = scene.preprocess(1)
    ts.append(time.perf_counter() - t0)
ts.sort()
print(wl, "preprocess(): median %.3f ms, min %.3f ms, visible %d" % (ts[len(ts) // 2] * 1e3, ts[0] * 1e3, pre.points.shape[0]))
