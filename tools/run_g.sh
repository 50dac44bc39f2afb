#!/bin/bash
# scratch: kernel stats of the depth sort alone: run_g.sh "N MODE [KEPT]" ...
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3g; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for cfg in "$@"; do
  rm -rf $O/p; rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o p -- python3 $R/tools/sort_probe.py $cfg > $O/log.txt 2>&1
  tail -1 $O/log.txt
  python3 - <<PY
import csv,glob
f=glob.glob("$O/p/**/*kernel_stats.csv", recursive=True)[0]
tot=0
for r in csv.DictReader(open(f)):
    if "gsx" in r["Name"]:
        nm=r["Name"].split("(anonymous namespace)::")[-1][:70]
        print("   %-70s calls %4s avg %8.1f us" % (nm, r["Calls"], float(r["AverageNs"])/1e3)); tot+=float(r["AverageNs"])/1e3
print("   sum %.1f us" % tot)
PY
done
