#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for r in 1048576 65536; do
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_rows_$r -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_anerf.py --steps 2 --rows-per-chunk $r > $GRAFT_REPO_ROOT/gpurun_out/prof_rows_$r.log 2>&1
python3 - <<PY
import csv,os
rows=list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof_rows_$r/p_kernel_stats.csv")))
print("rows per chunk $r")
for x in rows[:6]:
    print(x["Name"][:60].ljust(60), x["Calls"].rjust(6), "avg_us", round(float(x["AverageNs"])/1e3,1), "total_ms", round(float(x["TotalDurationNs"])/1e6,1))
PY
done
