#!/bin/bash
# usage: pmc_train.sh TAG : HBM bytes per launch of the training step's kernels (FETCH_SIZE / WRITE_SIZE passes, see pmc_hbm.sh)
TAG=$1
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmctrain_$TAG
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o $c -- python3 $GRAFT_REPO_ROOT/tools/bench_train.py --steps 4 --warmup 2 --no-graph > $OUT.$c.log 2>&1
done
python3 - <<PY
import csv, glob, os, collections
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmctrain_$TAG"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{root}/{c}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[r["Kernel_Name"].split("(")[0]][c].append(float(r["Counter_Value"]))
rows = []
for k, d in acc.items():
    fe = sum(d["FETCH_SIZE"]) / max(len(d["FETCH_SIZE"]), 1)
    wr = sum(d["WRITE_SIZE"]) / max(len(d["WRITE_SIZE"]), 1)
    rows.append((2 * fe + wr, k, len(d["FETCH_SIZE"]), fe, wr))
for t, k, n, fe, wr in sorted(rows, reverse=True)[:14]:
    print(k[:58].ljust(58), str(n).rjust(4), "read MB", round(2 * fe / 1024, 1), "write MB", round(wr / 1024, 1))
PY
