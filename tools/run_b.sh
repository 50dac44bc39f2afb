#!/bin/bash
# scratch: rocprofv3 kernel stats of one workload (test library, optional knobs from the environment) into gpurun_out/$1
TAG=$1; W=$2; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python3 $R/bench.py --workload $W --steps 20 --warmup 3 --repeats 3 --no-cpu-baseline --streams 1 "$@" > $O/prof.log 2>&1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/prof
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats.csv")))
for r in rows[:28]:
    print("%-90s calls %6s avg %10.1f ns  pct %s" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
PY
tail -c 600 $O/prof.log
