"""Frames per second with each frame replayed as ONE hipGraph launch (GaussianScene.capture_frame)
against the normal ~30 launches per frame, 3 frames in flight, alternating A/B in one process."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_scene, WORKLOADS

for wl in sys.argv[1:] or ["c2", "c3"]:
    n, w, h, _ = WORKLOADS[wl]
    sc, scene = build_scene(wl, "cuda:0")
    nflight = 3
    streams = [torch.cuda.Stream() for _ in range(nflight)]
    frames = [scene.capture_frame(1) for _ in range(nflight)]
    outs = [torch.empty((w, h, 3), device="cuda:0") for _ in range(nflight)]
    K = 100
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            with torch.cuda.stream(streams[i % nflight]):
                frames[i % nflight].replay()
        torch.cuda.synchronize()
        g = (time.perf_counter() - t0) / K * 1e3
        t0 = time.perf_counter()
        for i in range(K):
            with torch.cuda.stream(streams[i % nflight]):
                scene.render_image_hip(1, out=outs[i % nflight], no_sync=True)
        torch.cuda.synchronize()
        l = (time.perf_counter() - t0) / K * 1e3
        scene.confirm_frames()
        print("%s rep %d: graphs %.4f ms/frame, separate launches %.4f ms/frame" % (wl, rep, g, l))
    for f in frames:
        f.confirm()
