#!/bin/bash
# A/B of two builds of the test library on a rank's strip (bench.py --workload c4 --strip-of 8): bash tools/ab_strip.sh <tag> <old lib>
TAG=$1; OLD=$2
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for rep in 1 2; do
for side in new old; do
  if [ $side = old ]; then export GSX_TEST_LIB_PATH=$R/$OLD; else unset GSX_TEST_LIB_PATH; fi
  python bench.py --workload c4 --strip-of 8 --test-lib --repeats 10 > $O/strip_$side.json 2> $O/strip_$side.err
  python - $O/strip_$side.json $side <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("strip c4/8 %-4s frame %.4f ms  stages %s" % (sys.argv[2], d["frame_ms"]["median"], d["stage_ms"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
done
