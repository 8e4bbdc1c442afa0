#!/bin/bash
# gpurun -- 'bash profiles/trace_train.sh <tag>': BASELINE config 5 stand-in — one epoch of the fn training loop on bf16 GEMM
# operands (200 synthetic PU1K-shaped batches of 4 clouds x 64 patches x 12 points), f32 beside it, then the kernel table of a
# short bf16 epoch under rocprofv3.
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
mkdir -p $R/gpurun_out
python3 $R/profiles/train_step_microbench.py --batches 4 --epoch 200 --amp > $R/gpurun_out/${TAG}_train_epoch.jsonl
python3 $R/profiles/train_step_microbench.py --batches 4 --epoch 200 >> $R/gpurun_out/${TAG}_train_epoch.jsonl
python3 $R/profiles/train_step_microbench.py --batches 4,32 --steps 20 --amp >> $R/gpurun_out/${TAG}_train_epoch.jsonl
python3 $R/profiles/train_step_microbench.py --batches 4,32 --steps 20 >> $R/gpurun_out/${TAG}_train_epoch.jsonl
cat $R/gpurun_out/${TAG}_train_epoch.jsonl
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_train
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_train -- python3 $R/profiles/train_step_microbench.py --batches 4 --epoch 20 --amp > $R/gpurun_out/trace_train.log 2>&1
cd $R
python3 profiles/make_summary.py gpurun_out/trace_train "$TAG: rocprofv3 --kernel-trace --stats -- python3 profiles/train_step_microbench.py --batches 4 --epoch 20 --amp (fn training, bf16 GEMM operands, 23 steps incl. warm-up)" > gpurun_out/${TAG}_train_bf16_summary.md
find gpurun_out/trace_train -type f ! -name '*stats.csv' -delete
head -24 gpurun_out/${TAG}_train_bf16_summary.md
