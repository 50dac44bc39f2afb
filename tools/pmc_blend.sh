#!/bin/bash
# SQ counters of the compositing launch for one build of the test library:
#   bash tools/pmc_blend.sh <tag> <label> <workload> [path of libgsx_test.so to load instead of the tree's]
TAG=$1; LABEL=$2; W=$3; LIB=$4
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
if [ -n "$LIB" ]; then export GSX_TEST_LIB_PATH=$R/$LIB; else unset GSX_TEST_LIB_PATH; fi
cd /tmp && export TMPDIR=/tmp
P="python3 $R/bench.py --workload $W --steps 5 --warmup 1 --repeats 1 --no-cpu-baseline --streams 1 --test-lib"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_a_$LABEL -o p -- $P > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $O/pmc_b_$LABEL -o p -- $P > /dev/null 2>&1
for k in a b; do cp $(find $O/pmc_${k}_$LABEL -name "*counter_collection.csv" | head -1) $O/pmc_${k}_${LABEL}_$W.csv; rm -rf $O/pmc_${k}_$LABEL; done
python3 - $O/pmc_a_${LABEL}_$W.csv $O/pmc_b_${LABEL}_$W.csv $LABEL <<'PY'
import csv, sys, collections
for path in sys.argv[1:3]:
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if "blend_tile16" not in r["Kernel_Name"]: continue
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    print(sys.argv[3], " ".join("%s=%.4g" % (k, v[0] / max(v[1], 1)) for k, v in sorted(acc.items())))
PY
