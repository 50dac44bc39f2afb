#!/bin/bash
# kernel traces of two heavy-tailed workloads, one frame in flight, camera at rest (gpurun -- 'bash tools/attic/prof_two.sh'):
# what do the two compositing launches take, frame by frame?   python tools/attic/prof_two_read.py afterwards
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for w in ${@:-c3_clustered c3_trainedlike}; do
  rm -rf $R/gpurun_out/prof_$w
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$w -o $w -- python3 $R/bench.py --workload $w --steps 20 --warmup 3 --repeats 5 --no-cpu-baseline --streams 1 --camera-path none > $R/gpurun_out/prof_$w.log 2>&1
done
