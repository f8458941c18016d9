#!/bin/bash
# usage: prof.sh TAG  -> runs tests (quick subset) + rocprof bench, prints per-kernel table
TAG=$1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o $TAG -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-dense --no-sweep > $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_bench.log 2>&1
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"achieved": [0-9.]*\|"avg_launch_ms": [0-9.]*' $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_bench.log | tr '\n' ' '; echo
python3 - <<PY
import csv,os
rows=list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof_$TAG/${TAG}_kernel_stats.csv")))
# frames of the run (settle phase + warm-up + timed blocks + the reference frame): the final composite runs once per frame
frames = max([int(r["Calls"]) for r in rows if "k_composite_merged" in r["Name"]] + [1])
for r in rows[:10]:
    print(r["Name"][:42].ljust(42), r["Calls"].rjust(4), "avg_us", str(round(float(r["AverageNs"])/1e3,1)).rjust(8), "ms/frame", str(round(int(r["TotalDurationNs"])/1e6/frames,3)).rjust(7), r["Percentage"])
print("frames", frames, " sum of all kernels per frame ms", round(sum(int(r["TotalDurationNs"]) for r in rows)/1e6/frames,3),
      " without K3 (k_pe_mlp32 / k_pe_mlp16):", round(sum(int(r["TotalDurationNs"]) for r in rows if "k_pe_mlp" not in r["Name"])/1e6/frames,3))
PY
