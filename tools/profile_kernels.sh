#!/bin/bash
# rocprofv3 kernel-trace stats of one-frame-in-flight bench runs: bash tools/profile_kernels.sh <tag> [workloads...]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -o $w -- python3 $R/bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --streams 1 --camera-path none > $O/prof_$w.log 2>&1
  f=$(find $O/prof_$w -name "*kernel_stats.csv" | head -1)
  cp $f $O/${w}_kernel_stats.csv
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
calls = max(int(r["Calls"]) for r in rows if "blend" in r["Name"])
tot = 0.0
for r in rows:
    per = int(r["Calls"]) / calls
    if per < 0.5: continue
    us = float(r["AverageNs"]) / 1e3
    tot += us * per
    print("%-60s %5.1f x %8.1f us = %8.1f" % (r["Name"].split("(")[0][-60:], per, us, us * per))
print("sum per frame %.1f us" % tot)
PY
done
