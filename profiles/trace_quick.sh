set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace2 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/trace2.log 2>&1
cd $R
python3 profiles/make_summary.py gpurun_out/trace2 "fused" gpurun_out/fused_launches.json | head -12
python3 - <<PY
import json
d=json.load(open("gpurun_out/fused_launches.json"))
for k,v in d.items():
    if "<7" in k or "<0" in k: print(k, v[:12])
PY
rm -rf gpurun_out/trace2
