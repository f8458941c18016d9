#!/bin/bash
# usage: pmc_hbm.sh TAG : HBM bytes per launch of every kernel of a bench frame.
# Two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only, as MI355X_MICROARCH.md prescribes;
# hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (the gfx950 FETCH_SIZE x2 correction).  Writes gpurun_out/pmc_hbm_TAG.json
TAG=$1
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmchbm_$TAG
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o $c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dense --no-sweep > $OUT.$c.log 2>&1
done
python3 - <<PY
import csv, glob, hashlib, json, os, collections
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmchbm_$TAG"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{root}/{c}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[r["Kernel_Name"].split("(")[0]][c].append(float(r["Counter_Value"]))
out = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py --steps 3 --warmup 1; hbm_bytes = "
               "(2*FETCH_SIZE + WRITE_SIZE)*1024 with the gfx950 FETCH_SIZE x2 correction of MI355X_MICROARCH.md; averages over "
               "the coarse and fine launches of a frame", "kernels": {}}
# bench.py only quotes these numbers for the kernel sources they were measured on
h = hashlib.sha256()
for f in ("k_mlp32.hip", "mlp32_regs.inc", "k_mlp16.hip", "mlp16_core.hpp", "common.hpp"):      # = bench.py K3_SOURCES
    h.update(open(os.environ["GRAFT_REPO_ROOT"] + "/danbo-pytorch_amd/csrc/" + f, "rb").read())
out["kernel_src_sha16"] = h.hexdigest()[:16]
for k, d in acc.items():
    if not k.startswith("danbo::") and "danbo" not in k:
        continue
    fe = sum(d["FETCH_SIZE"]) / max(len(d["FETCH_SIZE"]), 1)
    wr = sum(d["WRITE_SIZE"]) / max(len(d["WRITE_SIZE"]), 1)
    out["kernels"][k.replace("void ", "")] = {"FETCH_SIZE_KB_avg": round(fe, 1), "WRITE_SIZE_KB_avg": round(wr, 1),
                                              "launches": len(d["FETCH_SIZE"]), "hbm_bytes_per_launch": int((2 * fe + wr) * 1024)}
json.dump(out, open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_hbm_$TAG.json", "w"), indent=1)
for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:12]:
    print(k[:60].ljust(60), v["launches"], round(v["hbm_bytes_per_launch"] / 1e6, 1), "MB/launch")
PY
