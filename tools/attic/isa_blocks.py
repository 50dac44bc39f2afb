"""Per basic block instruction mix of one kernel: python tools/attic/isa_blocks.py gsx_blend.hip 'blend_tile16_kernelILi1E' [min_exp]"""
import os, re, subprocess, sys
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "intro_to_gaussian_splatting_amd", "csrc")
src, pat = sys.argv[1], sys.argv[2]
min_exp = int(sys.argv[3]) if len(sys.argv) > 3 else 2
out = "/tmp/_isa_%s.s" % os.path.basename(src)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I../../include", "-I.", "-ffp-contract=off",
                "-fhip-fp32-correctly-rounded-divide-sqrt", "-fvisibility=hidden", "-S", "--cuda-device-only", "-o", out, src] + sys.argv[4:],
               cwd=csrc, check=True, capture_output=True)
s = open(out).read()
start = re.search(r"^(\S*%s\S*):.*\n" % re.escape(pat), s, re.M)
end = s.index("s_endpgm", start.end())
body = s[start.end():end]
blocks = re.split(r"\n(?=\.LBB\d+_\d+:)", body)
print(start.group(1)[:100], len(blocks), "blocks")
for b in blocks:
    lab = b.split("\n", 1)[0]
    ins = [l.strip() for l in b.split("\n") if l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;")]
    n_exp = sum(1 for l in ins if l.startswith("v_exp_f32"))
    n_scr = sum(1 for l in ins if l.startswith("scratch_"))
    if n_exp >= min_exp or n_scr:
        c = lambda p: sum(1 for l in ins if l.startswith(p))  # noqa: E731
        valu = sum(1 for l in ins if l.startswith("v_"))
        print("%-12s insts %4d valu %4d exp %2d pk_fma %2d pk_mul %2d pk_add %2d fma %2d mul %2d other_valu %3d ds_read %2d scratch %2d s_waitcnt %2d" % (
            lab[:12], len(ins), valu, n_exp, c("v_pk_fma"), c("v_pk_mul"), c("v_pk_add"), c("v_fma_f32") + c("v_fmac_f32"), c("v_mul_f32"),
            valu - n_exp - c("v_pk_fma") - c("v_pk_mul") - c("v_pk_add") - c("v_fma_f32") - c("v_fmac_f32") - c("v_mul_f32"), c("ds_read"), n_scr, c("s_waitcnt")))
