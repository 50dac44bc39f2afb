#!/bin/bash
# A/B of two builds of the test library on ONE GPU box: bash tools/ab_bench.sh <tag> <old libgsx_test.so> [workloads...]
# prints per workload the single-frame median and the per-stage times of both builds (bench.py --test-lib).
TAG=$1; OLD=$2; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for w in "$@"; do
  for side in new old; do
    if [ $side = old ]; then export GSX_TEST_LIB_PATH=$R/$OLD; else unset GSX_TEST_LIB_PATH; fi
    python bench.py --workload $w --test-lib --no-cpu-baseline --repeats 10 > $O/ab_${w}_$side.json 2> $O/ab_${w}_$side.err
    python - $O/ab_${w}_$side.json $w $side <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-14s %-4s frame %.4f ms  stages %s" % (sys.argv[2], sys.argv[3], d["frame_ms"]["median"], d["stage_ms"]))
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
  done
done
