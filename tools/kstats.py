"""Per-frame kernel table from a rocprofv3 kernel_stats CSV: python tools/kstats.py file.csv [...]"""
import csv, re, sys
for f in sys.argv[1:]:
    rows = list(csv.DictReader(open(f)))
    calls = max(int(r["Calls"]) for r in rows if "blend" in r["Name"])
    tot = nb = 0.0
    for r in rows:
        per = int(r["Calls"]) / calls
        if per < 0.5:
            continue
        us = float(r["AverageNs"]) / 1e3
        tot += us * per
        name = re.sub(r"\(anonymous namespace\)::|gsx::|void ", "", r["Name"])
        name = re.sub(r"\(.*", "", name)
        if "blend" not in name:
            nb += us * per
        print("%-58s %4.1f x %7.1f = %7.1f" % (name[:58], per, us, us * per))
    print("%s: sum %.1f us per frame, non-blend %.1f" % (f, tot, nb))
