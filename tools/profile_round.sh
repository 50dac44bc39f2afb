#!/bin/bash
# Evidence for one round, run ON the GPU box (gpurun -- 'bash tools/profile_round.sh r2'):
#   rocprofv3 kernel-trace stats of bench.py (one frame in flight) at C3, C2 and on the clustered scene,
#   three separate PMC passes at C3 (FETCH_SIZE; WRITE_SIZE; SQ_* + GRBM_GUI_ACTIVE; never combined with
#   other trace domains), the bench line of every workload, of the std_3dgs rule set and of a degree-3 SH
#   scene, the smoke.  Then, back in the build container:  python tools/summarize_profiles.py <tag>
TAG=${1:-r2}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in c3 c2 c3_clustered; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -o $w -- python3 $R/bench.py --workload $w --steps 20 --warmup 3 --repeats 5 --no-cpu-baseline --streams 1 > $O/prof_$w.log 2>&1
  cp $(find $O/prof_$w -name "*kernel_stats.csv" | head -1) $O/${w}_kernel_stats.csv
done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o c3 -- python3 $R/bench.py --steps 5 --warmup 1 --repeats 1 --no-cpu-baseline --streams 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o c3 -- python3 $R/bench.py --steps 5 --warmup 1 --repeats 1 --no-cpu-baseline --streams 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -o c3 -- python3 $R/bench.py --steps 5 --warmup 1 --repeats 1 --no-cpu-baseline --streams 1 > /dev/null 2>&1
for k in fetch write sq; do cp $(find $O/pmc_$k -name "*counter_collection.csv" | head -1) $O/pmc_$k.csv; done
cd $R
for w in c1 c2 c3 c4 c3_clustered; do python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err; done
python bench.py --semantics std_3dgs > $O/bench_c3_std3dgs.json 2> $O/bench_c3_std3dgs.err
python tools/make_synthetic_ply.py /tmp/c3_deg3.ply 1000000 3 > /dev/null 2>&1
python bench.py --workload c3 --ply /tmp/c3_deg3.ply > $O/bench_c3_sh3.json 2> $O/bench_c3_sh3.err
python __graft_entry__.py smoke 2>&1 | tail -1
rm -rf $O/prof_c3 $O/prof_c2 $O/prof_c3_clustered $O/pmc_fetch $O/pmc_write $O/pmc_sq
ls $O
