import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
sc, scene = bench.build_scene("c3", "cuda:0")
for _ in range(3): scene.render_image_hip(1)
torch.cuda.synchronize()
out = torch.empty((1920,1080,3), device="cuda:0")
N=50
t0=time.perf_counter()
for _ in range(N): scene.render_image_hip(1, out=out, no_sync=True)
t1=time.perf_counter()
torch.cuda.synchronize()
t2=time.perf_counter()
print("host enqueue per frame us", (t1-t0)/N*1e6, "total per frame us", (t2-t0)/N*1e6, "confirm", scene.confirm_frames())
# graph capture attempt
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    scene.render_image_hip(1, out=out, no_sync=True)   # allocate per-stream workspace
torch.cuda.synchronize(); scene.confirm_frames()
try:
    with torch.cuda.graph(g, stream=s):
        scene.render_image_hip(1, out=out, no_sync=True)
    torch.cuda.synchronize()
    ref = scene.render_image_hip(1)
    out.zero_()
    g.replay(); torch.cuda.synchronize()
    print("graph replay equal:", torch.equal(out, ref), "confirm", scene.confirm_frames())
    t0=time.perf_counter()
    for _ in range(N): g.replay()
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print("graph: host per frame us", (t1-t0)/N*1e6, "total per frame us", (t2-t0)/N*1e6)
except Exception as e:
    print("graph capture failed:", repr(e)[:500])
