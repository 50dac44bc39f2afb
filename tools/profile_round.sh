#!/bin/bash
# Evidence for one round, run ON the GPU box (gpurun -- 'bash tools/profile_round.sh r1i'):
#   rocprofv3 kernel-trace stats of bench.py at C3 with one and with three frames in flight,
#   three separate PMC passes (FETCH_SIZE; WRITE_SIZE; SQ_* + GRBM_GUI_ACTIVE; never combined with
#   other trace domains), the bench line of every workload and of the std_3dgs rule set, the smoke.
# Then, back in the build container:  python tools/summarize_profiles.py <tag>
TAG=${1:-r1i}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o c3 -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --streams 1 > $O/prof_$TAG.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_s3 -o c3 -- python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline > $O/prof_${TAG}_s3.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc3_fetch -o c3 -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --streams 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc3_write -o c3 -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --streams 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc3_sq -o c3 -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --streams 1 > /dev/null 2>&1
cd $R
for w in c1 c2 c3 c4; do python bench.py --steps 60 --warmup 6 --workload $w > $O/bench_$w.json 2> $O/bench_$w.err; done
python bench.py --steps 60 --warmup 6 --semantics std_3dgs > $O/bench_c3_std3dgs.json 2> $O/bench_c3_std3dgs.err
python __graft_entry__.py smoke 2>&1 | tail -1
ls $O
