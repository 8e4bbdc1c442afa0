#!/bin/bash
# Run on the GPU box (gpurun -- 'bash profiles/run_profiles.sh <tag> [--tests]'): the kernel trace and the three PMC passes
# (separate runs, as MI355X_MICROARCH.md prescribes) of bench.py; condensed outputs land in gpurun_out/<tag>_* for copying
# into profiles/.  The PMC passes run the plain step only (--no-roofline --no-strong-leg), so every kernel's launch count
# is (warmup + steps) x its launches per step and the per-step HBM traffic is the sum over kernels.
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
mkdir -p $R/gpurun_out
if [ "$2" == "--tests" ]; then
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $R/gpurun_out/gpu_tests.log 2>&1 || { tail -30 $R/gpurun_out/gpu_tests.log; exit 1; }
tail -3 $R/gpurun_out/gpu_tests.log
fi
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace $R/gpurun_out/pmc_f $R/gpurun_out/pmc_w $R/gpurun_out/pmc_m
BENCH="python3 $R/bench.py --no-cpu-baseline --no-strong-leg --no-m100 --no-ref-default"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace -- $BENCH --steps 3 --warmup 1 > $R/gpurun_out/trace.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_f -- $BENCH --steps 2 --warmup 1 --no-roofline > $R/gpurun_out/pmc_f.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_w -- $BENCH --steps 2 --warmup 1 --no-roofline > $R/gpurun_out/pmc_w.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_m -- $BENCH --steps 2 --warmup 1 --no-roofline > $R/gpurun_out/pmc_m.log 2>&1
cd $R
python3 profiles/make_summary.py gpurun_out/trace "$TAG: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg --no-m100 (1 x MI355X)" gpurun_out/${TAG}_gemm_launches.json > gpurun_out/${TAG}_summary.md
python3 profiles/pmc_to_json.py gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/${TAG}_pmc.json gpurun_out/pmc_m 3
cp gpurun_out/trace/*/*_kernel_stats.csv gpurun_out/${TAG}_kernel_stats.csv
# keep the merged-back payload small
find gpurun_out/trace gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/pmc_m -type f ! -name '*stats.csv' -delete
