#!/bin/bash
# On the GPU box: HBM fetch/write bytes and L2 hit/miss counts of the bias ring GEMM (r=1179648, k=n=512), default
# (interleaved) tile order against the contiguous one; one rocprofv3 --pmc pass per counter.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for lib in default CONTIG; do   # CONTIG: bash profiles/ablate.sh CONTIG
  for ctr in FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum; do
    rm -rf $R/gpurun_out/pg_${lib}_$ctr
    if [ $lib = CONTIG ]; then export SAPCU_LIB=$R/profiles/abl/libsapcu_CONTIG.so; else unset SAPCU_LIB; fi
    timeout -k 10 120 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/pg_${lib}_$ctr -- python3 $R/profiles/gemm_microbench.py 1179648 512 512 ring > $R/gpurun_out/pg.log 2>&1
    python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/pg_${lib}_$ctr/*/*_counter_collection.csv")[0]
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "gemm_ring" in r["Kernel_Name"] and r["Counter_Name"]=="$ctr"]
print("$lib $ctr launches=%d avg=%.4g"%(len(v), sum(v)/max(1,len(v))))
PY
    rm -rf $R/gpurun_out/pg_${lib}_$ctr
  done
done
