"""Register / scratch / occupancy / LDS table of one .hip file's kernels: python tools/kres.py gsx_blend.hip [-DGSX_TEST_HOOKS]"""
import os, re, subprocess, sys
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "intro_to_gaussian_splatting_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I../../include", "-I.", "-ffp-contract=off",
       "-fhip-fp32-correctly-rounded-divide-sqrt", "-fvisibility=hidden", "-Rpass-analysis=kernel-resource-usage", "-c", sys.argv[1],
       "-o", "/dev/null"] + sys.argv[2:]
out = subprocess.run(cmd, cwd=csrc, capture_output=True, text=True).stderr
name = None
row = {}
for ln in out.splitlines():
    m = re.search(r"remark: .*?(Function Name|VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", ln)
    if not m:
        if "error" in ln: print(ln)
        continue
    k, v = m.groups()
    if k == "Function Name":
        name = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().replace("gsx::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        row = {}
    else:
        row[k.split(" ")[0]] = v
        if k.startswith("LDS"):
            print("%-52s VGPR %3s AGPR %3s SGPR %3s scratch %4s occ %s LDS %6s" % (name[:52], row.get("VGPRs"), row.get("AGPRs"), row.get("SGPRs"), row.get("ScratchSize"), row.get("Occupancy"), row.get("LDS")))
