#!/bin/bash
# usage: pmc_gather.sh TAG : the transform / gather stage's HBM traffic as COUNTER readings (north_star: "rocprof HBM GB/s on the
# transform/gather stage"; VERDICT r5 weak 10: the round-4 figure was algorithmic bytes / time).  Two separate rocprofv3 --pmc passes
# (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only over tools/micro_gather.py, as MI355X_MICROARCH.md prescribes; per launch
# hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (the gfx950 FETCH_SIZE x2 correction), duration from the same runs' kernel
# traces.  Writes gpurun_out/gather_hbm_TAG.json (copy it to profiles/).
TAG=$1
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcgather_$TAG
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o $c -- python3 $GRAFT_REPO_ROOT/tools/micro_gather.py > $OUT.$c.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/micro_gather.py > $OUT.plain.log 2>&1
python3 - <<PY
import csv, glob, json, os, collections
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmcgather_$TAG"
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{root}/{c}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                cnt[r["Kernel_Name"].split("(")[0].replace("void ", "")][c].append(float(r["Counter_Value"]))
    for f in glob.glob(f"{root}/{c}/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"].split("(")[0].replace("void ", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
plain = [json.loads(l) for l in open(root + ".plain.log") if l.startswith("{")]
out = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over tools/micro_gather.py on the bench frame "
               "(512 x 512 x 48 coarse samples): hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch, the gfx950 correction of "
               "MI355X_MICROARCH.md; GB/s = counter bytes / the launch's duration UNDER THE PROFILER (kernel trace of the same pass); the "
               "un-profiled timings of the same script: unprofiled", "kernels": {}, "unprofiled": plain}
for k in sorted(cnt):
    if "k_bone_cull" not in k and "k_bone_gather" not in k:
        continue
    d = cnt[k]
    # the script launches the gather on 2 000 000 dense rows and on the compacted list: group launches by their write size
    groups = collections.defaultdict(list)
    for i, w in enumerate(d["WRITE_SIZE"]):
        groups[round(w, -4)].append(i)
    fe_all, us_all = d["FETCH_SIZE"], dur[k]
    for key, idx in sorted(groups.items()):
        wr = sum(d["WRITE_SIZE"][i] for i in idx) / len(idx)
        fe = sum(fe_all[i] for i in idx if i < len(fe_all)) / max(1, sum(1 for i in idx if i < len(fe_all)))
        us = sorted(us_all[i] for i in idx if i < len(us_all))
        med = us[len(us) // 2] if us else None
        b = (2 * fe + wr) * 1024
        out["kernels"].setdefault(k, []).append({"launches": len(idx), "FETCH_SIZE_KB": round(fe, 1), "WRITE_SIZE_KB": round(wr, 1),
                                                 "hbm_bytes_per_launch": int(b), "median_us_profiled": med,
                                                 "hbm_GBps": None if not med else round(b / med / 1e3, 1),
                                                 "frac_of_8TBps": None if not med else round(b / med / 1e3 / 8000, 3)})
json.dump(out, open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/gather_hbm_$TAG.json", "w"), indent=1)
print(json.dumps(out["kernels"], indent=1)[:3000])
PY
