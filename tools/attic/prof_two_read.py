"""Per-kernel medians over the LAST 40 frames of the traces tools/attic/prof_two.sh left under gpurun_out/."""
import csv, glob, re, sys, collections
import numpy as np
for w in sys.argv[1:] or ["c3_clustered", "c3_trainedlike"]:
    f = glob.glob("gpurun_out/prof_%s/**/*kernel_trace.csv" % w, recursive=True)
    if not f:
        print(w, "no trace"); continue
    rows = [r for r in csv.DictReader(open(f[0])) if "gsx" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    per = collections.defaultdict(list)
    for r in rows:
        m = re.search(r"(\w+_kernel(<[^>]*>)?)", r["Kernel_Name"])
        name = m.group(1) if m else r["Kernel_Name"][:30]
        per[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
    nframes = len(per.get("blend_tile16_kernel<1>", [])) or 1
    print(w, "frames", nframes)
    tot = 0.0
    for name, d in sorted(per.items(), key=lambda kv: -np.median(kv[1][-40 * max(1, len(kv[1]) // nframes):]) * max(1, len(kv[1]) // nframes)):
        k = max(1, round(len(d) / nframes))
        tail = d[-40 * k:]
        print("  %-34s x%d  median %7.1f us  (p10 %7.1f p90 %7.1f)" % (name[:34], k, np.median(tail), np.percentile(tail, 10), np.percentile(tail, 90)))
        tot += np.median(tail) * k
    print("  sum of medians %.1f us" % tot)
