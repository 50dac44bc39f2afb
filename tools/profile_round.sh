#!/bin/bash
# Evidence for one round, run ON the GPU box (gpurun -- 'bash tools/profile_round.sh r3'):
#   rocprofv3 kernel-trace stats of bench.py (one frame in flight) at C3, C2, C4, on the clustered scene and on a 1/8
#   strip of C4 (what one rank of BASELINE config 5 renders; also from spatially ordered rows: strip_spatial); three separate PMC passes (FETCH_SIZE; WRITE_SIZE; SQ_* +
#   GRBM_GUI_ACTIVE; never combined with other trace domains) at C3, C2, C4 and on the strip; the bench line of every
#   workload, of the std_3dgs rule set, of a degree-3 SH scene and of the reference's notebook workload; the smoke.
#   Then, back in the build container:  python tools/summarize_profiles.py <tag>
TAG=${1:-r6}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 20 --warmup 3 --repeats 5 --no-cpu-baseline --streams 1 --camera-path none"
P="python3 $R/bench.py --steps 5 --warmup 1 --repeats 1 --no-cpu-baseline --streams 1 --camera-path none"
for w in c3 c2 c4 c3_clustered c3_trainedlike strip strip_spatial; do
  if [ $w = strip ]; then A="--workload c4 --strip-of 8"; elif [ $w = strip_spatial ]; then A="--workload c4 --strip-of 8 --spatial-order"; else A="--workload $w"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -o $w -- $B $A > $O/prof_$w.log 2>&1
  cp $(find $O/prof_$w -name "*kernel_stats.csv" | head -1) $O/${w}_kernel_stats.csv
  rm -rf $O/prof_$w
done
for w in c3 c2 c4 strip strip_spatial; do
  if [ $w = strip ]; then A="--workload c4 --strip-of 8"; elif [ $w = strip_spatial ]; then A="--workload c4 --strip-of 8 --spatial-order"; else A="--workload $w"; fi
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$w -o p -- $P $A > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$w -o p -- $P $A > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq_$w -o p -- $P $A > /dev/null 2>&1
  # (the raw counter files -- one row per dispatch, counter and hardware instance -- run to tens of MB; what the summary
  # uses is the per-kernel distribution: keep the columns it reads and about 40 rows per kernel and counter)
  for k in fetch write sq; do
    python3 - $(find $O/pmc_${k}_$w -name "*counter_collection.csv" | head -1) $O/pmc_${k}_$w.csv <<'PY'
import collections, csv, sys
rows = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "gsx" in r["Kernel_Name"]:
        rows[(r["Kernel_Name"], r["Counter_Name"])].append(r["Counter_Value"])
with open(sys.argv[2], "w", newline="") as f:
    wr = csv.writer(f)
    wr.writerow(["Kernel_Name", "Counter_Name", "Counter_Value"])
    for (kn, cn), vals in rows.items():
        step = max(1, -(-len(vals) // 40))
        for v in vals[::step]:
            wr.writerow([kn, cn, v])
PY
    rm -rf $O/pmc_${k}_$w
  done
done
cd $R
for w in c1 c2 c3 c4 c3_clustered c3_trainedlike c3_1m2 notebook; do python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err; done
python bench.py --workload c4 --strip-of 8 > $O/bench_strip.json 2> $O/bench_strip.err
python bench.py --workload c4 --strip-of 8 --spatial-order > $O/bench_strip_spatial.json 2> $O/bench_strip_spatial.err
python bench.py --semantics std_3dgs > $O/bench_c3_std3dgs.json 2> $O/bench_c3_std3dgs.err
python tools/make_synthetic_ply.py /tmp/c3_deg3.ply 1000000 3 > /dev/null 2>&1
python bench.py --workload c3 --ply /tmp/c3_deg3.ply > $O/bench_c3_sh3.json 2> $O/bench_c3_sh3.err
python __graft_entry__.py smoke 2>&1 | tail -1
du -sh $O; ls -la $O | sort -k5 -n | tail -5
