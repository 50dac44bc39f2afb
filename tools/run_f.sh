#!/bin/bash
# scratch: SQ counters of the compositing kernel under two libraries
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3f; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for lib in "" "_r2"; do
  export GSX_TEST_LIB_PATH=$R/intro_to_gaussian_splatting_amd/libgsx_test$lib.so
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/t$lib -o t -- python3 $R/bench.py --test-lib --workload $1 --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --streams 1 > $O/log$lib.txt 2>&1
  f=$(find $O/t$lib -name "*counter_collection.csv" | head -1)
  python3 - <<PY
import csv, collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open("$f")):
    if "blend_tile16" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("lib '$lib':", {k: "%.4g" % (sorted(v)[len(v)//2]) for k,v in sorted(acc.items())})
PY
  rm -rf $O/t$lib
done
