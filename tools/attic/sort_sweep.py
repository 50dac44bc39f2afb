import subprocess, sys
code = '''
import torch, ctypes, sys
sys.path.insert(0, ".")
from intro_to_gaussian_splatting_amd import _ffi
lib = _ffi.load_test_hooks()
n, bits, key16 = %d, %d, %d
gen = torch.Generator(device="cuda:0").manual_seed(n)
hi = (1 << bits) - 1
keys64 = torch.randint(0, min(hi, 5000) + 1, (n,), generator=gen, device="cuda:0", dtype=torch.int64)
if bits > 16: keys64 = keys64 * 65537 %% (hi + 1)
vals = torch.arange(n, device="cuda:0", dtype=torch.int32)
scratch = torch.empty(lib.gsx_workspace_bytes(n, 16, 16, 16, n), dtype=torch.uint8, device="cuda:0")
res = []
for live in (n, max(1, (2 * n) // 3), max(1, n // 5)):
    keys = keys64.to(torch.int16 if key16 else torch.int32).clone()
    v = vals.clone()
    count = torch.tensor([live], dtype=torch.int32, device="cuda:0")
    rc = lib.gsx_debug_sort_pairs(keys.data_ptr(), v.data_ptr(), n, bits, key16, count.data_ptr() if live != n else None, scratch.data_ptr(), scratch.numel(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ref_k, ref_p = torch.sort(keys64[:live], stable=True)
    ok = torch.equal(keys[:live].to(torch.int64) & (0xFFFF if key16 else 0xFFFFFFFF), ref_k) and torch.equal(v[:live].to(torch.int64), ref_p)
    res.append((live, rc, ok))
    print("partial", res, flush=True)
print("all", res)
'''
for n, bits, k16 in [(524289,16,1),(524289,32,0),(524289,8,1),(600000,16,1),(532480,16,1),(540672,16,1),(557056,16,1),(1000000,16,1),(1000000,13,1),(4219511,13,1),(5300000,13,1)]:
    r = subprocess.run([sys.executable, "-c", code % (n, bits, k16)], capture_output=True, text=True)
    out = (r.stdout.strip().splitlines() or ["-"])[-1]
    print(n, bits, k16, "->", out if r.returncode == 0 else "CRASH rc=%d %s" % (r.returncode, (r.stderr.strip().splitlines() or [""])[-1][:100]))
