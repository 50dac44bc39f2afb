#!/bin/bash
# scratch: kernel stats of any python tool: run_h.sh TAG script args...
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf $O/p; rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o p -- python3 $R/$@ > $O/log.txt 2>&1
tail -2 $O/log.txt
python3 - <<PY
import csv,glob
f=glob.glob("$O/p/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "gsx" in r["Name"]:
        nm=r["Name"].split("(anonymous namespace)::")[-1][:60]
        print("   %-60s calls %5s avg %8.1f us" % (nm, r["Calls"], float(r["AverageNs"])/1e3))
PY
