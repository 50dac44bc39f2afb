"""Does a replayed captured frame clear the never-rendered border of its output buffer?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_scene

sc, scene = build_scene("c1", "cuda:0")
ref = scene.render_image_hip(1).clone()
a = scene.capture_frame(1, headroom=2.0)
print("after capture: equal", torch.equal(a.out, ref))
a.out.fill_(7.0)
a.replay(); torch.cuda.synchronize()
bad = (a.out != ref)
print("after fill+replay: equal", torch.equal(a.out, ref), "bad pixels", int(bad.any(dim=2).sum()),
      "bad x range", bad.any(dim=2).any(dim=1).nonzero().flatten()[:3].tolist(), bad.any(dim=2).any(dim=1).nonzero().flatten()[-3:].tolist())
out = torch.empty_like(ref); out.fill_(7.0)
scene.render_image_hip(1, out=out, no_sync=True); torch.cuda.synchronize(); scene.confirm_frames()
print("plain no_sync call into a dirty buffer: equal", torch.equal(out, ref))
s1 = torch.cuda.Stream()
a.out.fill_(7.0); torch.cuda.synchronize()
with torch.cuda.stream(s1):
    a.replay()
torch.cuda.synchronize()
print("replay on another stream: equal", torch.equal(a.out, ref))
