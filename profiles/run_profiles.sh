#!/bin/bash
# Run on the GPU box (gpurun -- 'bash profiles/run_profiles.sh [--no-tests]'): GPU tests, then the kernel trace and the three
# PMC passes (separate runs) of bench.py; condensed outputs land in gpurun_out/r01_* for copying into profiles/.
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
if [ "$1" != "--no-tests" ]; then
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $R/gpurun_out/gpu_tests.log 2>&1 || { tail -30 $R/gpurun_out/gpu_tests.log; exit 1; }
tail -3 $R/gpurun_out/gpu_tests.log
fi
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace $R/gpurun_out/pmc_f $R/gpurun_out/pmc_w $R/gpurun_out/pmc_m
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/trace.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_f -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_f.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_w -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_w.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_m -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_m.log 2>&1
cd $R
python3 profiles/make_summary.py gpurun_out/trace "round 1: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline (1 x MI355X)" gpurun_out/r01_gemm_launches.json > gpurun_out/r01_summary.md
python3 profiles/pmc_to_json.py gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/r01_pmc.json gpurun_out/pmc_m
cp gpurun_out/trace/*/*_kernel_stats.csv gpurun_out/r01_kernel_stats.csv
# keep the merged-back payload small
find gpurun_out/trace gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/pmc_m -type f ! -name '*stats.csv' -delete
