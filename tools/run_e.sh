#!/bin/bash
# scratch: kernel trace of the compositing kernel under two libraries (resource columns + durations)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3e; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for lib in "" "_r2"; do
  export GSX_TEST_LIB_PATH=$R/intro_to_gaussian_splatting_amd/libgsx_test$lib.so
  rocprofv3 --kernel-trace --output-format csv -d $O/t$lib -o t -- python3 $R/bench.py --test-lib --workload $1 --steps 5 --warmup 1 --repeats 1 --no-cpu-baseline --streams 1 > $O/log$lib.txt 2>&1
  f=$(find $O/t$lib -name "*kernel_trace.csv" | head -1)
  python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$f")) if "blend_tile16" in r["Kernel_Name"]]
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows]
r=rows[-1]
print("lib '$lib': n=%d  median %.1f us  min %.1f" % (len(d), sorted(d)[len(d)//2], min(d)), {k:r[k] for k in r if k in ("VGPR_Count","Accum_VGPR_Count","SGPR_Count","LDS_Block_Size","Scratch_Size","Workgroup_Size","Grid_Size","Private_Segment_Size")})
PY
  rm -rf $O/t$lib
done
