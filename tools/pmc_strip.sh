#!/bin/bash
# PMC passes of one rank's 1/8 strip of C4, rows as given / in Morton order with block bounds (run ON the GPU box):
#   gpurun -- 'bash tools/pmc_strip.sh r6'    -> gpurun_out/<tag>/pmc_strip_{plain,spatial}_{fetch,write,sq}.csv
TAG=${1:-r6}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P="python3 $R/bench.py --steps 5 --warmup 1 --repeats 1 --no-cpu-baseline --streams 1 --camera-path none --workload c4 --strip-of 8"
for so in plain spatial; do
  A=""; [ $so = spatial ] && A="--spatial-order"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/p_fetch_$so -o p -- $P $A > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/p_write_$so -o p -- $P $A > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p_sq_$so -o p -- $P $A > /dev/null 2>&1
  for k in fetch write sq; do
    python3 - $(find $O/p_${k}_$so -name "*counter_collection.csv" | head -1) $O/pmc_strip_${so}_${k}.csv <<'PY'
import collections, csv, sys
rows = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "gsx" in r["Kernel_Name"]:
        rows[(r["Kernel_Name"].split("(anonymous namespace)::")[-1].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(sys.argv[2], "w", newline="") as f:
    wr = csv.writer(f)
    wr.writerow(["Kernel_Name", "Counter_Name", "Dispatches", "Median_Counter_Value"])
    for (kn, cn), vals in sorted(rows.items()):
        vals.sort()
        wr.writerow([kn, cn, len(vals), vals[len(vals) // 2]])
PY
    rm -rf $O/p_${k}_$so
  done
done
cat $O/pmc_strip_*_fetch.csv $O/pmc_strip_*_write.csv | grep -i "project\|prepare"
cat $O/pmc_strip_*_sq.csv | grep -i "project_window"
