#!/bin/bash
# scratch: a subset of the GPU tests + bench lines for the compositing kernel
TAG=${1:-r3c}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -s -k "$2" > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -v "amdgpu.ids" $O/pytest.log | tail -${4:-25}
for w in $3; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline --repeats 10 > $O/bench_$w.json 2> $O/bench_$w.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$w.json").read().strip().splitlines()[-1])
    print("$w", d["value"], d["frame_ms"]["median"], d["stage_ms"], d["config"]["tile_instances"])
except Exception as e:
    print("$w failed", e); print(open("$O/bench_$w.err").read()[-800:])
PY
done
