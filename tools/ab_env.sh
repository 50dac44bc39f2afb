#!/bin/bash
# The same test library under two settings of a GSX_* knob, alternating: bash tools/ab_env.sh <tag> <workload> VAR=value [reps]
TAG=$1; W=$2; KV=$3; REPS=${4:-2}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for rep in $(seq $REPS); do
  for side in default knob; do
    if [ $side = knob ]; then export "$KV"; else unset ${KV%%=*}; fi
    python bench.py --workload $W --test-lib --no-cpu-baseline --repeats 10 --camera-path none > $O/env_${W}_$side.json 2> $O/env_${W}_$side.err
    python - $O/env_${W}_$side.json $W "$side ($KV)" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-14s %-28s frame %.4f ms  blend %.4f" % (sys.argv[2], sys.argv[3], d["frame_ms"]["median"], d["stage_ms"]["blend"]))
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
  done
done
