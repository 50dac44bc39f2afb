"""Turns the rocprofv3 outputs a gpurun call left under gpurun_out/ into the summaries committed
under profiles/ (kernel stats per frame, PMC traffic / VALU busy of the compositing kernel).

    python tools/summarize_profiles.py r1i          # reads gpurun_out/prof_<tag>*, pmc3_*, bench_c*.json
"""
import collections
import csv
import json
import re
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r1i"
pmc = sys.argv[2] if len(sys.argv) > 2 else "pmc3"


def short(name):
    m = re.search(r"(count_kernel<[\w ,]+>|row_scan_kernel|scatter_kernel<[\w ,]+>|scan_block_sums_kernel|"
                  r"scan_apply_kernel|blend_\w+|project_\w+|emit_kernel<[\w ]+>|tile_ranges_kernel<[\w ]+>|"
                  r"fillBuffer\w*|copyBuffer\w*|FillFunctor)", name)
    return m.group(1) if m else name[:40]


lines = []
for suffix, label in (("", "1 frame in flight"), ("_s3", "3 frames in flight")):
    rows = list(csv.DictReader(open("gpurun_out/prof_%s%s/c3_kernel_stats.csv" % (tag, suffix))))
    frames = [int(r["Calls"]) for r in rows if "blend_tile16" in r["Name"]][0]
    lines.append("## C3, %s (%d frames profiled)\n" % (label, frames))
    lines.append("| kernel | launches/frame | avg us | us/frame |\n|---|---|---|---|")
    tot = 0.0
    for r in rows:
        per = float(r["TotalDurationNs"]) / frames / 1e3
        tot += per
        if per >= 0.5:
            lines.append("| `%s` | %.1f | %.1f | %.1f |" % (short(r["Name"]), int(r["Calls"]) / frames,
                                                            float(r["AverageNs"]) / 1e3, per))
    lines.append("| **sum** | | | **%.1f** |\n" % tot)
    shutil.copy("gpurun_out/prof_%s%s/c3_kernel_stats.csv" % (tag, suffix),
                "profiles/%s_c3_kernel_stats_%s.csv" % (tag, "3streams" if suffix else "1stream"))

out = {"source": "rocprofv3 --pmc <counters> --kernel-trace (one pass per counter set) -- python3 bench.py --steps 5 "
                 "--warmup 1 --no-cpu-baseline --streams 1  (C3: 1M Gaussians, 1920x1080), MI355X, ROCm 7.2",
       "units": "FETCH_SIZE / WRITE_SIZE are KiB per dispatch (bytes = value*1024); on gfx950 FETCH_SIZE reports 1/2 of "
                "coalesced read bytes (MI355X_MICROARCH.md, HBM section), calibrated below on project_pack_kernel whose "
                "read set is exactly 56 B x 1e6 Gaussians; SQ_* cycle counters are quad-cycles summed over all SIMDs",
       "kernels": {}}


def k2(n):
    for key in ("blend_tile16_kernel", "project_pack_kernel", "emit_kernel", "tile_ranges_kernel",
                "scan_block_sums_kernel", "scan_apply_kernel", "row_scan_kernel"):
        if key in n:
            return key
    for key in ("count_kernel", "scatter_kernel"):
        if key in n:
            return key + ("<u16>" if "unsigned short" in n else "<u32>")
    return None


for kind in ("fetch", "write", "sq"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open("gpurun_out/%s_%s/c3_counter_collection.csv" % (pmc, kind))):
        k = k2(r["Kernel_Name"])
        if k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            if kind == "sq" and k not in ("blend_tile16_kernel", "project_pack_kernel"):
                continue
            out["kernels"].setdefault(k, {})[c] = round(sum(v) / len(v), 2)
b, p = out["kernels"]["blend_tile16_kernel"], out["kernels"]["project_pack_kernel"]
out["calibration"] = {"project_pack_read_bytes_expected": 56e6, "FETCH_SIZE_bytes": p["FETCH_SIZE"] * 1024,
                      "ratio": round(p["FETCH_SIZE"] * 1024 / 56e6, 4), "project_pack_write_bytes_expected": 68e6,
                      "WRITE_SIZE_bytes": p["WRITE_SIZE"] * 1024}
out["blend_traffic_bytes_per_launch"] = {"read_corrected_x2": b["FETCH_SIZE"] * 1024 * 2, "write": b["WRITE_SIZE"] * 1024,
                                         "total": b["FETCH_SIZE"] * 1024 * 2 + b["WRITE_SIZE"] * 1024}
cyc = b["GRBM_GUI_ACTIVE"] / 8.0
out["blend_valu"] = {"kernel_cycles_per_xcd": cyc, "valu_busy_frac": round(b["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc, 4),
                     "valu_instructions": b["SQ_INSTS_VALU"],
                     "cycles_per_valu_instruction": round(b["SQ_ACTIVE_INST_VALU"] * 4 / b["SQ_INSTS_VALU"], 3),
                     "note": "valu_busy = SQ_ACTIVE_INST_VALU*4 / (1024 SIMDs * GRBM_GUI_ACTIVE/8 XCDs)"}
json.dump(out, open("profiles/r1_pmc_c3.json", "w"), indent=1)

lines.append("## Bench lines (profiles/%s_bench_c*.json)\n" % tag)
lines.append("| config | ms/frame (3 in flight) | Mpixel/s | ms/frame (1 in flight) | blend ms | max abs dpixel | CPU port Mpix/s |\n"
             "|---|---|---|---|---|---|---|")
for w in ("c1", "c2", "c3", "c4"):
    shutil.copy("gpurun_out/bench_%s.json" % w, "profiles/%s_bench_%s.json" % (tag, w))
    d = json.load(open("gpurun_out/bench_%s.json" % w))
    lines.append("| %s | %.4f | %.0f | %.4f | %.4f | %.2g | %.2f |" % (
        w, d["ms_per_step"], d["value"], d["config"]["ms_per_frame_one_in_flight"], d["roofline"]["avg_ms"],
        d.get("max_abs_dpixel", float("nan")), d["cpu_baseline"]["value"]))
import os
if os.path.exists("gpurun_out/bench_c3_std3dgs.json"):
    shutil.copy("gpurun_out/bench_c3_std3dgs.json", "profiles/%s_bench_c3_std3dgs.json" % tag)
    d = json.load(open("gpurun_out/bench_c3_std3dgs.json"))
    lines.append("| c3, std_3dgs rules | %.4f | %.0f | %.4f | %.4f | %.2g (+ %d threshold flips <= %.2g) | %.2f |" % (
        d["ms_per_step"], d["value"], d["config"]["ms_per_frame_one_in_flight"], d["roofline"]["avg_ms"],
        d.get("max_abs_dpixel", float("nan")), d["cpu_baseline"]["threshold_flip_pixels"],
        d["cpu_baseline"]["max_abs_dpixel_incl_flips"], d["cpu_baseline"]["value"]))
lines.append("\n## PMC, compositing kernel (profiles/r1_pmc_c3.json)\n")
lines.append("```\n%s\n%s\n%s\n```" % (json.dumps(out["calibration"]), json.dumps(out["blend_traffic_bytes_per_launch"]),
                                        json.dumps(out["blend_valu"])))
open("profiles/%s_SUMMARY.md" % tag, "w").write("# rocprofv3 summary, round 1 (%s), MI355X\n\n" % tag + "\n".join(lines) + "\n")
print("\n".join(lines))
