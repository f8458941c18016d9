#!/bin/bash
# usage: pmc_anerf.sh TAG : HBM bytes of one A-NeRF frame (config 5), per kernel and in total.  Two separate rocprofv3 --pmc passes
# (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only; hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 correction, see pmc_hbm.sh)
TAG=$1
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcanerf_$TAG
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o $c -- python3 $GRAFT_REPO_ROOT/tools/bench_anerf.py --steps 1 --warmup 1 > $OUT.$c.log 2>&1
done
python3 - <<PY
import csv, glob, hashlib, json, os, collections
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmcanerf_$TAG"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for f in glob.glob(f"{root}/{c}/*counter_collection.csv"):
        rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    # two frames were rendered (one warm-up: weight packing, engine refresh, allocator fills; one steady-state): only the SECOND
    # is attributed -- a frame starts at its cylinder-bounds kernel
    starts = [i for i, r in enumerate(rows) if "k_cylinder_chunk" in r["Kernel_Name"] or "k_cylinder_pass1" in r["Kernel_Name"]]
    assert len(starts) >= 2, len(starts)
    for r in rows[starts[-1]:]:
        acc[r["Kernel_Name"].split("(")[0]][c].append(float(r["Counter_Value"]))
out = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/bench_anerf.py --steps 1 --warmup 1; attributed: "
               "the SECOND (steady-state) frame of 512 x 512 x (48 + 16) samples only; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
               "(gfx950 FETCH_SIZE x2 correction)", "kernels": {}}
h = hashlib.sha256()
for f in ("k_linear16.hip", "k_anerf.hip", "common.hpp"):
    h.update(open(os.environ["GRAFT_REPO_ROOT"] + "/danbo-pytorch_amd/csrc/" + f, "rb").read())
out["kernel_src_sha16"] = h.hexdigest()[:16]
total = 0
for k, d in acc.items():
    fe, wr = sum(d["FETCH_SIZE"]), sum(d["WRITE_SIZE"])
    b = int((2 * fe + wr) * 1024)
    total += b
    out["kernels"][k.replace("void ", "")] = {"launches": len(d["FETCH_SIZE"]), "hbm_bytes_per_frame": b,
                                              "hbm_bytes_per_launch": b // max(len(d["FETCH_SIZE"]), 1)}
out["frame_hbm_bytes"] = total
json.dump(out, open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_anerf_$TAG.json", "w"), indent=1)
for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_frame"])[:8]:
    print(k[:60].ljust(60), v["launches"], round(v["hbm_bytes_per_launch"] / 1e6, 1), "MB/launch", round(v["hbm_bytes_per_frame"] / 1e9, 2), "GB/frame")
print("frame total GB", round(total / 1e9, 2))
PY
