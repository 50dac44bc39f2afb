"""Turns the rocprofv3 outputs a `tools/profile_round.sh <tag>` call left under gpurun_out/<tag>/ into the
summaries committed under profiles/ (kernel stats per frame, HBM-side traffic of every kernel against its
algorithmic bytes, VALU busy of the compositing kernel, the bench lines).

    python tools/summarize_profiles.py r2
"""
import collections
import csv
import json
import os
import re
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
src = "gpurun_out/%s" % tag


def short(name):
    name = re.sub(r"\(anonymous namespace\)::|gsx::|void ", "", name)
    return re.sub(r"\(.*", "", name)[:70]


def kernel_table(path, label):
    rows = list(csv.DictReader(open(path)))
    frames = max(int(r["Calls"]) for r in rows if "blend" in r["Name"])
    lines = ["## %s (%d frames profiled, one frame in flight)\n" % (label, frames),
             "| kernel | launches/frame | avg us | us/frame |\n|---|---|---|---|"]
    tot = nonblend = 0.0
    for r in rows:
        per_frame = int(r["Calls"]) / frames
        if per_frame < 0.5:
            continue
        per = float(r["AverageNs"]) / 1e3 * per_frame
        tot += per
        if "blend" not in r["Name"]:
            nonblend += per
        lines.append("| `%s` | %.1f | %.1f | %.1f |" % (short(r["Name"]), per_frame, float(r["AverageNs"]) / 1e3, per))
    lines.append("| **sum** (non-compositing kernels: **%.1f**) | | | **%.1f** |\n" % (nonblend, tot))
    return lines, tot, nonblend


lines = []
for w, label in (("c3", "C3: 1M Gaussians, 1920x1080"), ("c2", "C2: 100k Gaussians, 1920x1080"),
                 ("c3_clustered", "clustered: 1M Gaussians, half of them in 5 % of the frame")):
    p = "%s/%s_kernel_stats.csv" % (src, w)
    if os.path.exists(p):
        shutil.copy(p, "profiles/%s_%s_kernel_stats_1stream.csv" % (tag, w))
        lines += kernel_table(p, label)[0]

# ---- PMC: HBM-side traffic of every kernel of a C3 frame against its algorithmic bytes
d3 = json.load(open("%s/bench_c3.json" % src))
N, D = d3["config"]["n_gaussians"], d3["config"]["tile_instances"]
npix = 119 * 67 * 256
ALG = {   # algorithmic bytes per launch at C3
    "project_pack_kernel": (56 * N, 60 * N),
    "count_kernel<u32>": (4 * N, 0.5e6),
    "row_scan_kernel": (0.5e6, 0.5e6),
    "scatter_kernel<u32,first>": (12 * N, 16 * N),     # keys + rectangles in; keys, values, rectangles out
    "scatter_kernel<u32>": (8 * N, 8 * N),
    "scatter_kernel<u32,final>": (16 * N, 12 * N),
    "sample_rank_kernel": (128 * 2048 * 4, 1024),     # every one of its 128 workgroups reads the 2048 sample keys
    "count_kernel<u32,split>": (4 * N, 0.5e6),
    "bucket_sort_kernel": (16 * N, 12 * N),
    "chunk_sums_kernel": (8 * N, 0),
    "emit_kernel": (12 * N, 6 * D),
    "count_kernel<u16>": (2 * D, 1.2e6),
    "scatter_kernel<u16>": (6 * D, 6 * D),
    "tile_ranges_kernel": (2 * D, 64e3),
    "tile_schedule_kernel": (64e3, 32e3),
    "blend_tile16_kernel": (40 * D, 12 * npix),
}


def k2(n):
    n = re.sub(r"\(anonymous namespace\)::|gsx::|void ", "", n)
    for key in ("blend_tile16_kernel", "project_pack_kernel", "emit_kernel", "tile_ranges_kernel", "chunk_sums_kernel",
                "row_scan_kernel", "sample_rank_kernel", "bucket_sort_kernel", "small_depth_sort_kernel",
                "tile_schedule_kernel"):
        if key in n:
            return key
    m = re.match(r"(count|scatter)_kernel<(unsigned short|unsigned int), (?:true|false|\d+), (\w+)", n)
    if m:
        kind, key, mode = m.group(1), "u16" if "short" in m.group(2) else "u32", m.group(3)
        if kind == "count" and n.rstrip(">").endswith("true") and key == "u32":
            return "count_kernel<u32,split>"
        if kind == "scatter" and mode == "1":
            return "scatter_kernel<u32,first>"
        if kind == "scatter" and mode == "2":
            return "scatter_kernel<u32,final>"
        return "%s_kernel<%s>" % (kind, key)
    return None


out = {"source": "rocprofv3 --pmc <counters> --kernel-trace (one pass per counter set) -- python3 bench.py --steps 5 --warmup 1 "
                 "--repeats 1 --no-cpu-baseline --streams 1  (C3: 1M Gaussians, 1920x1080), MI355X, ROCm 7.2",
       "units": "FETCH_SIZE / WRITE_SIZE are KiB per dispatch (bytes = value*1024); on gfx950 FETCH_SIZE reports 1/2 of coalesced "
                "read bytes (MI355X_MICROARCH.md, HBM section), re-calibrated below on project_pack_kernel whose read set is exactly "
                "56 B x 1e6 Gaussians; SQ_* cycle counters are quad-cycles summed over all SIMDs",
       "kernels": {}}
for kind in ("fetch", "write", "sq"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open("%s/pmc_%s.csv" % (src, kind))):
        k = k2(r["Kernel_Name"])
        if k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            if kind == "sq" and k not in ("blend_tile16_kernel", "project_pack_kernel"):
                continue
            out["kernels"].setdefault(k, {})[c] = round(sum(v) / len(v), 2)
p = out["kernels"]["project_pack_kernel"]
corr = 56.0 * N / (p["FETCH_SIZE"] * 1024)          # ~2 on gfx950
out["calibration"] = {"project_pack_read_bytes_expected": 56.0 * N, "FETCH_SIZE_bytes": p["FETCH_SIZE"] * 1024,
                      "fetch_correction": round(corr, 4), "project_pack_write_bytes_expected": 60.0 * N,
                      "WRITE_SIZE_bytes": p["WRITE_SIZE"] * 1024}
traffic = {}
lines.append("## HBM-side traffic per launch at C3 against algorithmic bytes (PMC; FETCH_SIZE x %.3f as calibrated)\n" % corr)
lines.append("| kernel | read MB (alg) | written MB (alg) | traffic / algorithmic |\n|---|---|---|---|")
for k, (ar, aw) in ALG.items():
    m = out["kernels"].get(k)
    if not m or "FETCH_SIZE" not in m or "WRITE_SIZE" not in m:
        continue
    rd, wr = m["FETCH_SIZE"] * 1024 * corr, m["WRITE_SIZE"] * 1024
    ratio = (rd + wr) / (ar + aw)
    traffic[k] = {"read_bytes": rd, "write_bytes": wr, "algorithmic_read": ar, "algorithmic_write": aw, "ratio": round(ratio, 3)}
    lines.append("| `%s` | %.1f (%.1f) | %.1f (%.1f) | %.2f |" % (k, rd / 1e6, ar / 1e6, wr / 1e6, aw / 1e6, ratio))
out["traffic_vs_algorithmic"] = traffic
b = out["kernels"]["blend_tile16_kernel"]
out["blend_traffic_bytes_per_launch"] = {"read_corrected": b["FETCH_SIZE"] * 1024 * corr, "write": b["WRITE_SIZE"] * 1024,
                                         "total": b["FETCH_SIZE"] * 1024 * corr + b["WRITE_SIZE"] * 1024}
cyc = b["GRBM_GUI_ACTIVE"] / 8.0
out["blend_valu"] = {"kernel_cycles_per_xcd": cyc, "valu_busy_frac": round(b["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc, 4),
                     "valu_instructions": b["SQ_INSTS_VALU"],
                     "cycles_per_valu_instruction": round(b["SQ_ACTIVE_INST_VALU"] * 4 / b["SQ_INSTS_VALU"], 3),
                     "note": "valu_busy = SQ_ACTIVE_INST_VALU*4 / (1024 SIMDs * GRBM_GUI_ACTIVE/8 XCDs)"}
json.dump(out, open("profiles/%s_pmc_c3.json" % tag, "w"), indent=1)

lines.append("\n## Bench lines (profiles/%s_bench_*.json)\n" % tag)
lines.append("| workload | Mpixel/s (value: median single frame) | ms/frame median [min, max] | ms/frame, 3 in flight | blend ms | "
             "max abs dpixel vs CPU port | parity_ok | CPU port Mpix/s |\n|---|---|---|---|---|---|---|---|")
for w in ("c1", "c2", "c3", "c4", "c3_clustered", "c3_std3dgs", "c3_sh3"):
    p = "%s/bench_%s.json" % (src, w)
    if not os.path.exists(p) or os.path.getsize(p) == 0:
        continue
    shutil.copy(p, "profiles/%s_bench_%s.json" % (tag, w))
    d = json.load(open(p))
    f = d["frame_ms"]
    lines.append("| %s | %.0f | %.4f [%.4f, %.4f] | %s | %.4f | %.2g | %s | %.2f |" % (
        w, d["value"], f["median"], f["min"], f["max"], d["config"].get("ms_per_frame_in_flight"), d["roofline"]["avg_ms"],
        d.get("max_abs_dpixel", float("nan")), d.get("parity_ok"), d.get("cpu_baseline", {}).get("value", float("nan"))))
lines.append("\n## PMC, compositing kernel (profiles/%s_pmc_c3.json)\n" % tag)
lines.append("```\n%s\n%s\n%s\n```" % (json.dumps(out["calibration"]), json.dumps(out["blend_traffic_bytes_per_launch"]),
                                        json.dumps(out["blend_valu"])))
open("profiles/%s_SUMMARY.md" % tag, "w").write("# rocprofv3 summary (%s), MI355X\n\n" % tag + "\n".join(lines) + "\n")
print("\n".join(lines))
