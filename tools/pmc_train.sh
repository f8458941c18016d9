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
import hashlib, json
STEPS = 6                     # 2 warm-up + 4 steps of bench_train.py, all profiled
rows, total = [], 0.0
out = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/bench_train.py --steps 4 --warmup 2 --no-graph "
               "(config 4: 3072 rays x (32 + 16)); hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE x2 correction); per "
               "training step = totals / 6", "kernels": {}}
for k, d in acc.items():
    fe = sum(d["FETCH_SIZE"]) / max(len(d["FETCH_SIZE"]), 1)
    wr = sum(d["WRITE_SIZE"]) / max(len(d["WRITE_SIZE"]), 1)
    rows.append((2 * fe + wr, k, len(d["FETCH_SIZE"]), fe, wr))
    b = (2 * sum(d["FETCH_SIZE"]) + sum(d["WRITE_SIZE"])) * 1024
    total += b
    out["kernels"][k.replace("void ", "")] = {"launches_per_step": round(len(d["FETCH_SIZE"]) / STEPS, 2), "hbm_bytes_per_launch": int((2 * fe + wr) * 1024),
                                              "hbm_bytes_per_step": int(b / STEPS)}
out["step_hbm_bytes"] = int(total / STEPS)
h = hashlib.sha256()
for f in ("k_train.hip", "k_linear16.hip", "k_dw16.hip", "k_assign_bwd.hip", "k_train_rows.hip", "common.hpp"):
    h.update(open(os.environ["GRAFT_REPO_ROOT"] + "/danbo-pytorch_amd/csrc/" + f, "rb").read())
out["kernel_src_sha16"] = h.hexdigest()[:16]
json.dump(out, open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_train_$TAG.json", "w"), indent=1)
for t, k, n, fe, wr in sorted(rows, reverse=True)[:14]:
    print(k[:58].ljust(58), str(n).rjust(4), "read MB", round(2 * fe / 1024, 1), "write MB", round(wr / 1024, 1))
print("HBM GB per step", round(out["step_hbm_bytes"] / 1e9, 2))
PY
