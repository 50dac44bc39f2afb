#!/bin/bash
# A fuzz campaign on the GPU box: tools/fuzz.py over every profile, one summary line each.
#   bash tools/fuzz_campaign.sh <first seed> [seeds per default profile] [guard] > gpurun_out/fuzz_campaign.txt
# guard: every frame once more inside buffers with guarded margins (tools/fuzz.py)
S=${1:-20000}; N=${2:-400}; X=${3:-}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { # label, first, count, extra args
  local label=$1 first=$2 count=$3; shift 3
  local t0=$(date +%s)
  out=$(timeout 3000 python tools/fuzz.py $first $count "$@" $X 2>&1 | grep -v amdgpu.ids | tail -1)
  echo "$label $X [$first, $((first + count))) $(( $(date +%s) - t0 )) s: $out"
}
run default $S $N
run big $((S + 10000)) $((N / 4)) big
run extreme $((S + 20000)) $((N / 2)) extreme
run plain $((S + 30000)) $((N / 2)) plain
for fam in trained needle tie few mixed; do
  run family=$fam $((S + 40000)) $((N / 4)) family=$fam
  S=$((S + 1000))
done
