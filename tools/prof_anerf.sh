#!/bin/bash
# usage: prof_anerf.sh TAG -> rocprof kernel stats of the A-NeRF frame bench (config 5)
TAG=$1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o $TAG -- python3 $GRAFT_REPO_ROOT/tools/bench_anerf.py > $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}.log | cut -c1-300
python3 - <<PY
import csv,os
rows=list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof_$TAG/${TAG}_kernel_stats.csv")))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[:12]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(5), "avg_us", str(round(float(r["AverageNs"])/1e3,1)).rjust(8), r["Percentage"])
PY
