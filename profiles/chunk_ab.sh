#!/bin/bash
# fn in patch chunks sized to the 256 MB Infinity Cache (SAPCU_CHUNK, read at model create) against the one-chunk
# form: step time per chunk size, and FETCH_SIZE of the d = 512 chain kernel at chunk 512 (q|k|v of a chunk = 151 MB) and 4096.
# Run on the GPU box: gpurun -- 'bash profiles/chunk_ab.sh'  -> gpurun_out/r04_chunk_ab.txt
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_chunk_ab.txt
: > $OUT
BENCH="python3 $R/bench.py --no-cpu-baseline --no-strong-leg --no-m100 --no-ref-default --no-roofline"
for c in 0 2048 1024 512 256; do
    if [ $c == 0 ]; then unset SAPCU_CHUNK; else export SAPCU_CHUNK=$c; fi
    $BENCH --steps 8 --warmup 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunk %s: %.3f ms per 4096-query step' % ('$c' if '$c' != '0' else '4096 (default)', d['ms_per_step']))" >> $OUT
done
cd /tmp && export TMPDIR=/tmp
for c in 0 512; do
    if [ $c == 0 ]; then unset SAPCU_CHUNK; else export SAPCU_CHUNK=$c; fi
    rm -rf $R/gpurun_out/pmc_chunk
    timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_chunk -- $BENCH --steps 2 --warmup 1 > /dev/null 2>&1
    python3 - $R/gpurun_out/pmc_chunk $c >> $OUT <<'P'
import csv, glob, sys, collections
tot = collections.Counter()
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            k = r["Kernel_Name"].split("(")[0].replace("void sapcu::", "").replace("sapcu::", "")
            tot[k] += float(r["Counter_Value"]) * 1024.0 * 2.0        # KiB, x 2 on gfx950 (profiles/pmc_to_json.py)
steps = 3.0
label = sys.argv[2] if sys.argv[2] != "0" else "4096"
print("chunk %s: fetched per 4096-query step, all kernels: %.2f GB" % (label, sum(tot.values()) / steps / 1e9))
for k in sorted(tot, key=tot.get, reverse=True)[:4]:
    print("chunk %s:   %-42s %.2f GB per step" % (label, k, tot[k] / steps / 1e9))
P
done
cat $OUT
