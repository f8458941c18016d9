#!/bin/bash
# usage: prof_train.sh TAG [bench_train args] -> rocprof kernel stats of 20 training steps (config 4)
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o $TAG -- python3 $GRAFT_REPO_ROOT/tools/bench_train.py --steps 20 --warmup 5 "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}.log | cut -c1-200
python3 - <<PY
import csv,os
rows=list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof_$TAG/${TAG}_kernel_stats.csv")))
for r in rows[:28]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(5), "avg_us", str(round(float(r["AverageNs"])/1e3,1)).rjust(8), "ms/step", str(round(int(r["TotalDurationNs"])/25e6,3)).rjust(7), r["Percentage"])
print("kernels per step", sum(int(r["Calls"]) for r in rows)/25, "GPU ms per step", round(sum(int(r["TotalDurationNs"]) for r in rows)/25e6,3))
PY
