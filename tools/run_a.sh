#!/bin/bash
# scratch driver for one gpurun call: tests + a few bench lines into gpurun_out/$1
TAG=${1:-r3a}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -k "${2:-test}" > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
for w in c3 c4 c3_1m2 c2; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline --repeats 10 > $O/bench_$w.json 2> $O/bench_$w.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$w.json").read().strip().splitlines()[-1])
    print("$w", d["value"], d["frame_ms"]["median"], d["stage_ms"], d["config"]["tile_instances"])
except Exception as e:
    print("$w failed", e)
PY
done
timeout 600 python bench.py --workload notebook > $O/bench_notebook.json 2> $O/bench_notebook.err; tail -c 1500 $O/bench_notebook.json; tail -3 $O/bench_notebook.err
