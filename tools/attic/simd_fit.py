"""What does a SIMD's finishing time depend on?  One converged frame of a bench workload on libgsx_test.so with the blend
probe on; least squares of every SIMD's last end on what its waves did: entries walked on whole trips, entries walked
under the exact rule, batches staged, waves.   python tools/attic/simd_fit.py [workload]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from intro_to_gaussian_splatting_amd import _ffi
_ffi.use_test_library()
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c3_trainedlike"
sc, scene = bench.build_scene(wl, "cuda")
lib = _ffi.load()
for _ in range(12):
    scene.render_image_hip(1)
torch.cuda.synchronize()
half = 1 << 17
buf = torch.zeros((4 * half, 4), dtype=torch.int32, device="cuda")
lib.gsx_debug_set_blend_probe(buf.data_ptr())
st = {}
scene.render_image_hip(1, stats=st, timing=True)
torch.cuda.synchronize()
lib.gsx_debug_set_blend_probe(None)
d = buf.cpu().numpy().view(np.uint32)
rows = np.nonzero(d[:half, 2] > 0)[0]
sec = d[half + rows]
helper = (d[rows, 1] & 0x40000000) != 0
walked = (d[rows, 3] & 0x7FFFFFFF).astype(np.float64)
batches = np.where(helper, np.ceil(d[rows, 2] / 64.0), (sec[:, 0] & 0xFFF).astype(np.float64))
after = np.where(helper, 0.0, (sec[:, 0] >> 12).astype(np.float64))
start, end = sec[:, 3].astype(np.int64), sec[:, 1].astype(np.int64)
t0 = start.min()
end = (end - t0) * 0.01
simd = ((sec[:, 2] >> 16).astype(np.int64) << 16) | (sec[:, 2] & 0xFF30)
keys, inv = np.unique(simd, return_inverse=True)
S = lambda v: np.bincount(inv, weights=v)     # noqa: E731
last = np.zeros(len(keys)); np.maximum.at(last, inv, end)
X = np.stack([np.ones(len(keys)), S(walked - after), S(after), S(batches)], axis=1)
coef, *_ = np.linalg.lstsq(X, last, rcond=None)
pred = X @ coef
print(wl, "blend %.3f ms, span %.1f us, %d SIMDs" % (st["stage_ms"]["blend"], end.max(), len(keys)))
print("last end [us] = %.1f + %.4f x entries on whole trips + %.4f x entries under the exact rule + %.3f x batches   (R^2 %.2f)" % (
    coef[0], coef[1], coef[2], coef[3], 1 - ((last - pred) ** 2).sum() / ((last - last.mean()) ** 2).sum()))
print("in entries on whole trips: exact rule x %.2f, a batch = %.1f" % (coef[2] / coef[1], coef[3] / coef[1]))
old = S(walked + 5 * batches)
print("corr(last end, walked + 5 batches) %.2f; corr(last end, fitted) %.2f; per SIMD: walked %.0f (exact %.0f), batches %.0f" % (
    np.corrcoef(last, old)[0, 1], np.corrcoef(last, pred)[0, 1], S(walked).mean(), S(after).mean(), S(batches).mean()))
