"""Register / scratch / occupancy budget of the hot kernels, from the compiler's own report (hipcc cross-compiles: no GPU).

Round 6 lost 9 % of the 4K frame to a two-line change that kept a 64-bit word live in the 64-VGPR compositing instance:
32 more bytes of scratch per lane, no test noticed, a profile did -- three GPU runs later.  The numbers below are what
the tree was measured with; a change that moves one of them is to be re-measured (bench.py --workload c4 / c3) and the
number updated with the measurement in the commit message."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "intro_to_gaussian_splatting_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I../../include", "-I.", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fvisibility=hidden", "-Rpass-analysis=kernel-resource-usage", "-c"]

# kernel (demangled prefix) -> (max VGPRs, max scratch bytes per lane, min waves per SIMD)
BUDGET = {
    "gsx_blend.hip": {
        "blend_tile16_ref_kernel": (97, 0, 4),       # the instance a frame runs (evaluates reference-order records)
        "blend_tile16_kernel<1>": (64, 96, 8),       # GSX_FLAG_PLAIN_FOOTPRINTS: 8 waves per SIMD, 96 B spilled (C4: 1.51 ms)
        "blend_generic_kernel": (72, 0, 7),
    },
    "gsx_project.hip": {
        "project_pack_kernel<false, -1>": (46, 0, 7),
        "project_window_kernel<false, -1>": (73, 0, 6),
        "project_stage_kernel": (34, 0, 8),
    },
}


def _resources(src):
    out = subprocess.run([HIPCC] + FLAGS + [src, "-o", "/dev/null"], cwd=CSRC, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    table, name, row = {}, None, {}
    for ln in out.stderr.splitlines():
        m = re.search(r"remark: .*?(Function Name|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", ln)
        if not m:
            continue
        k, v = m.groups()
        if k == "Function Name":
            name = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
            name = name.replace("gsx::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            row = {}
        else:
            row[k.split(" ")[0]] = int(v)
            if k.startswith("LDS"):
                table[name] = dict(row)
    return table


@pytest.mark.skipif(not os.path.exists(HIPCC) or shutil.which("c++filt") is None, reason="needs hipcc and c++filt")
@pytest.mark.parametrize("src", sorted(BUDGET))
def test_hot_kernels_stay_inside_their_register_budget(src):
    table = _resources(src)
    for kernel, (vgprs, scratch, occupancy) in BUDGET[src].items():
        assert kernel in table, (kernel, sorted(table))
        r = table[kernel]
        assert r["VGPRs"] <= vgprs and r["ScratchSize"] <= scratch and r["Occupancy"] >= occupancy, (kernel, r)
