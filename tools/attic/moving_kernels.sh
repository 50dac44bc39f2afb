# Per-kernel times of one movable-camera hipGraph with the camera at rest and moving (tools/attic/moving_probe.py under rocprofv3):
#   bash tools/attic/moving_kernels.sh WORKLOAD [...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for w in "$@"; do for m in rest moving; do
  rm -rf /tmp/mv; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mv -o o -- python3 $R/tools/attic/moving_probe.py $w $m > /tmp/mv.log 2>&1 || tail -3 /tmp/mv.log
  echo "== $w $m"; python3 $R/tools/kstats.py $(find /tmp/mv -name "*kernel_stats.csv" | head -1) | cut -c1-100
done; done
