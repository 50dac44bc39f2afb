"""Time of the SH -> RGB kernel alone (1M Gaussians, degree 3: 216 MB of algorithmic traffic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes
from intro_to_gaussian_splatting_amd import _ffi
lib = _ffi.load()
n = 1_000_000
for deg in (0, 1, 2, 3):
    k = (deg + 1) ** 2
    pts = torch.randn((n, 3), device="cuda:0"); sh = torch.randn((n, k, 3), device="cuda:0")
    out = torch.empty((n, 3), device="cuda:0")
    c = (ctypes.c_float * 3)(0.1, 0.2, 5.0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    call = lambda: lib.gsx_sh_to_rgb(ctypes.c_void_p(pts.data_ptr()), ctypes.c_void_p(sh.data_ptr()), deg, n, c,
                                     ctypes.c_void_p(out.data_ptr()), st)
    for _ in range(5): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    bytes_ = n * (12 + 12 * k + 12)
    print("degree %d: %.1f us, %.0f MB -> %.2f TB/s (%.0f %% of 8 TB/s)" % (deg, us, bytes_ / 1e6, bytes_ / us / 1e6, bytes_ / us / 1e6 / 8 * 100))
