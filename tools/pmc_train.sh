#!/bin/bash
# usage: pmc_train.sh TAG : HBM bytes of ONE steady-state training step (config 4), per kernel and in total.
# Two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only, as MI355X_MICROARCH.md prescribes;
# hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (the gfx950 FETCH_SIZE x2 correction).  The process runs 3 warm-up + 4 timed
# steps (no HIP graph: the counters are per dispatch); only the LAST THREE steps are attributed -- a step = the dispatches from
# one k_trunk_prep (first kernel after the step's zeroing) up to the next -- so weight packing of the first call, allocator
# fills and lazy initialisations are not counted.
TAG=$1
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmctrain_$TAG
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o $c -- python3 $GRAFT_REPO_ROOT/tools/bench_train.py --steps 4 --warmup 3 --no-graph > $OUT.$c.log 2>&1
done
python3 - <<PY
import csv, glob, os, collections, hashlib, json
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmctrain_$TAG"
per = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for f in glob.glob(f"{root}/{c}/*counter_collection.csv"):
        rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    starts = [i for i, r in enumerate(rows) if "k_trunk_prep" in r["Kernel_Name"]]
    assert len(starts) >= 5, len(starts)
    seg = rows[starts[-4]:starts[-1]]                  # the last three complete steps
    acc = collections.defaultdict(list)
    for r in seg:
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    per[c] = acc
STEPS = 3
out = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/bench_train.py --steps 4 --warmup 3 --no-graph "
               "(config 4: 3072 rays x (32 + 16)); hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE x2 correction); "
               "attributed: the last three complete steps only (k_trunk_prep to k_trunk_prep), divided by 3", "kernels": {}}
total = 0.0
rows = []
for k in sorted(set(per["FETCH_SIZE"]) | set(per["WRITE_SIZE"])):
    fe, wr = per["FETCH_SIZE"].get(k, []), per["WRITE_SIZE"].get(k, [])
    b = (2 * sum(fe) + sum(wr)) * 1024
    total += b
    n = max(len(fe), len(wr))
    out["kernels"][k] = {"launches_per_step": round(n / STEPS, 2), "hbm_bytes_per_launch": int(b / max(n, 1)), "hbm_bytes_per_step": int(b / STEPS)}
    rows.append((b / STEPS, k, n / STEPS, 2 * sum(fe) * 1024 / STEPS, sum(wr) * 1024 / STEPS))
out["step_hbm_bytes"] = int(total / STEPS)
h = hashlib.sha256()
for f in ("k_train.hip", "k_mlp16.hip", "k_mlp16_bwd.hip", "mlp16_core.hpp", "k_dw16.hip", "k_assign_bwd.hip", "k_train_rows.hip", "k_train_head.hip",
          "common.hpp"):
    h.update(open(os.environ["GRAFT_REPO_ROOT"] + "/danbo-pytorch_amd/csrc/" + f, "rb").read())
out["kernel_src_sha16"] = h.hexdigest()[:16]
json.dump(out, open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_train_$TAG.json", "w"), indent=1)
for t, k, n, fe, wr in sorted(rows, reverse=True)[:14]:
    print(k[:58].ljust(58), str(round(n, 1)).rjust(5), "read MB", round(fe / 1e6, 1), "write MB", round(wr / 1e6, 1))
print("HBM GB per step", round(out["step_hbm_bytes"] / 1e9, 3))
PY
