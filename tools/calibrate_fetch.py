"""FETCH_SIZE against known footprints (tools/microbench_gather.hip): python tools/calibrate_fetch.py <counter csv> <program stdout>
Prints, per configuration and launch, FETCH_SIZE bytes (KiB x 1024, uncorrected) and the ratio footprint / FETCH_SIZE for the
64-byte and 128-byte footprints -- the correction factor that access pattern needs."""
import csv
import json
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == "FETCH_SIZE"]
cfgs = [ln.split() for ln in open(sys.argv[2]) if ln.startswith("CFG ")]
g = [float(r["Counter_Value"]) * 1024 for r in rows if "gather_kernel" in r["Kernel_Name"]]
st = [float(r["Counter_Value"]) * 1024 for r in rows if "stream_kernel" in r["Kernel_Name"]]
out = {}
gi = 0
for c in cfgs:
    name = c[1]
    kv = {c[i]: c[i + 1] for i in range(2, len(c) - 1, 2) if not c[i + 1].startswith("(")}
    if name.startswith("stream"):
        b = float(c[c.index("bytes") + 1])
        out[name] = {"bytes": b, "FETCH_SIZE_bytes": st[0], "correction": round(b / st[0], 4)}
        continue
    b64, b128, lst = float(c[c.index("bytes64") + 1]), float(c[c.index("bytes128") + 1]), float(c[c.index("list_bytes") + 1])
    for label in ("cold", "warm"):
        f = g[gi]
        gi += 1
        out["%s/%s" % (name, label)] = {"footprint_64B": b64 + lst, "footprint_128B": b128 + lst, "FETCH_SIZE_bytes": f,
                                        "correction_if_64B_sectors": round((b64 + lst) / f, 4),
                                        "correction_if_128B_lines": round((b128 + lst) / f, 4)}
print(json.dumps(out, indent=1))
