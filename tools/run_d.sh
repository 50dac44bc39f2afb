#!/bin/bash
# scratch: A/B of knobs on the test library: run_d.sh TAG "workloads" "ENV1=.. ENV2=..|ENV..." (configs separated by |)
TAG=$1; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
IFS='|' read -ra CFGS <<< "$3"
for rep in 1 2; do
for w in $2; do
  for cfg in "${CFGS[@]}"; do
    env $cfg python bench.py --test-lib --workload $w --no-cpu-baseline --repeats 8 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']
print('%-14s %-40s median %.4f  blend %.4f  sort %.4f scan %.4f bin %.4f proj %.4f' % ('$w', '$cfg', d['frame_ms']['median'], s['blend'], s['depth_sort'], s['scan'], s['bin'], s['project']))"
  done
done
done
