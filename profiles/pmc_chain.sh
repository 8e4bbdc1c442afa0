#!/bin/bash
# gpurun -- 'bash profiles/pmc_chain.sh': L2-miss bytes (FETCH_SIZE, raw 32-byte... units as rocprofv3 reports them) of the three chain kernels
# launched alone by profiles/chain_ab.py, with the real neighbour tables and with every neighbour = point 0 (gathers always hit):
# separates the weight stream's refetches from the k / v gathers' misses.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for z in 0 1; do
  rm -rf $R/gpurun_out/pmc_chain_$z
  if [ $z == 1 ]; then export SAPCU_AB_ZERO_IDX=1; fi
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_chain_$z -- python3 $R/profiles/chain_ab.py > $R/gpurun_out/pmc_chain_$z.log 2>&1
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/pmc_chain_$z/*/*counter_collection.csv")[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "fn_edge_chain" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        agg[r["Kernel_Name"].split("(")[0][-40:]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print("zero_idx=$z", k, "launches", len(v), "FETCH_SIZE per launch (raw units)", sum(v) / len(v))
PY
  find $R/gpurun_out/pmc_chain_$z -type f -delete
done
