"""Per-kernel times of the depth sort alone (gsx_debug_depth_sort in libgsx_test.so) -- run under rocprofv3:
    rocprofv3 --kernel-trace --stats ... -- python3 tools/attic/sort_probe.py N MODE [KEPT_FRACTION]
MODE: 0 LSD, 1 = 256 buckets, 2 = 1024 buckets, 4 = LSD with the rectangles carried along, -1 = the route of
gsx_render_forward."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from intro_to_gaussian_splatting_amd import _ffi
lib = _ffi.load_test_hooks()
n, mode = int(sys.argv[1]), int(sys.argv[2])
kept = float(sys.argv[3]) if len(sys.argv) > 3 else 0.95
rs = np.random.RandomState(1)
keys = rs.uniform(0.2, 40.0, n).astype(np.float32).view(np.uint32).copy()
keys[rs.uniform(size=n) > kept] = 0xFFFFFFFE
rect = rs.randint(0, 100, size=(n, 4)).astype(np.uint16)
d_keys0 = torch.from_numpy(keys.view(np.int32).copy()).cuda()
d_rect = torch.from_numpy(rect.view(np.int16).copy()).cuda()
d_rrect = torch.zeros_like(d_rect)
d_order = torch.zeros(n, dtype=torch.int32, device="cuda")
nbytes = 24 * n + 8192 + lib.gsx_workspace_bytes(n, 16, 16, 16, 1)
scratch = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
counts = (ctypes.c_int64 * 3)()
hint = int((keys < 0xFFFFFFFE).sum()) if kept < 0.9 else 0
for it in range(12):
    d_keys = d_keys0.clone()
    rc = lib.gsx_debug_depth_sort(d_keys.data_ptr(), n, d_rect.data_ptr(), d_rrect.data_ptr(), d_order.data_ptr(), mode, 0, hint,
                                  counts, scratch.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
print("n", n, "mode", mode, "route", counts[2], "kept", counts[0])
