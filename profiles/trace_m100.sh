#!/bin/bash
# gpurun -- 'bash profiles/trace_m100.sh': kernel trace of the step at the reference's default patch size (M = 100)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_m100
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_m100 -- python3 $R/profiles/m100_check.py > $R/gpurun_out/trace_m100.log 2>&1
cd $R
python3 profiles/make_summary.py gpurun_out/trace_m100 "M=100, batch 400 (fused passes of 4000 seeds), 40 000 seeds + warm-up" > gpurun_out/m100_summary.md
find gpurun_out/trace_m100 -type f ! -name '*stats.csv' -delete
tail -2 gpurun_out/trace_m100.log; head -28 gpurun_out/m100_summary.md
