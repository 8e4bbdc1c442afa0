#!/bin/bash
# Per-kernel same-box A/B: the kernel trace of bench.py's plain step under two library builds (SAPCU_LIB_PATH), average
# microseconds per launch of the kernels whose name matches a pattern.
#   bash profiles/kernel_ab.sh <libA.so|default> <libB.so|default> <name pattern (grep -E)> <out.txt>
R=$GRAFT_REPO_ROOT
A=$1; B=$2; PAT=$3; OUT=$(realpath -m ${4:-$R/gpurun_out/kernel_ab.txt})
: > $OUT
cd /tmp && export TMPDIR=/tmp
for L in "$A" "$B"; do
    if [ "$L" == "default" ]; then unset SAPCU_LIB_PATH; else export SAPCU_LIB_PATH=$R/$L; fi
    rm -rf $R/gpurun_out/trace_ab
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_ab -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-strong-leg --no-m100 --no-ref-default --no-roofline > /dev/null 2>&1
    echo "== $L" >> $OUT
    python3 - $R/gpurun_out/trace_ab "$PAT" >> $OUT <<'P'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
tot = 0.0
for r in csv.DictReader(open(f)):
    if re.search(sys.argv[2], r["Name"]):
        n, t = int(r["Calls"]), float(r["TotalDurationNs"])
        tot += t
        print("  %-60s %4d calls  %9.1f us avg" % (r["Name"][:60], n, t / n / 1e3))
print("  matching kernels, total per profiled run: %.3f ms" % (tot / 1e6))
P
done
rm -rf $R/gpurun_out/trace_ab
cat $OUT
