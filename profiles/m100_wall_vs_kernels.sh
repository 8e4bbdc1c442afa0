#!/bin/bash
# gpurun -- 'bash profiles/m100_wall_vs_kernels.sh': wall time per refine() at M = 48 / 100 (profiles/sync_check.py) against the sum of the
# kernel durations of the same process (rocprofv3 --kernel-trace): shows what is not kernel time
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_sc
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_sc -- python3 $R/profiles/sync_check.py > $R/gpurun_out/trace_sc.log 2>&1
grep "per refine" $R/gpurun_out/trace_sc.log
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/trace_sc/*/*_kernel_trace.csv")[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
# the last 5 refine() passes at M = 100 are the tail of the trace: find them by the knn_outer launches (one per refine)
starts = [i for i, r in enumerate(rows) if "knn_outer" in r[2]]
for name, sl in (("M=48 (passes 3-7)", starts[2:7] + [starts[7]]), ("M=100 (last 5)", starts[-5:] + [len(rows)])):
    for a, b in zip(sl[:-1], sl[1:]):
        seg = rows[a:b]
        busy = sum(e - s for s, e, _ in seg) / 1e6
        span = (seg[-1][1] - seg[0][0]) / 1e6
        gaps = sorted(((seg[i + 1][0] - seg[i][1]) / 1e3, seg[i][2][:40], seg[i + 1][2][:40]) for i in range(len(seg) - 1))[-3:]
        print(name, "kernels %.2f ms, first start to last end %.2f ms, %d launches; largest gaps (us):" % (busy, span, len(seg)), [(round(g, 1), a_, b_) for g, a_, b_ in gaps])
PY
find $R/gpurun_out/trace_sc -type f -delete
