#!/bin/bash
# usage: pmc_mlp32.sh TAG : SQ counter passes over tools/micro_mlp32.py (k_pe_mlp16 and k_pe_mlp32 on the same rows; each --pmc set in its own run)
TAG=$1
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- python3 $GRAFT_REPO_ROOT/tools/micro_mlp32.py 3 > $OUT.p$i.log 2>&1
  tail -1 $OUT.p$i.log
done
python3 - <<PY
import csv, glob, os, collections
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_$TAG"
for kernel in ("k_pe_mlp16", "k_pe_mlp32"):
    vals = collections.defaultdict(list)
    for f in glob.glob(root + "/p*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = []
    for f in glob.glob(root + "/p*/*kernel_trace.csv"):
        dur += [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(f)) if kernel in r["Kernel_Name"]]
    big = [d for d in dur if d > 0.2 * max(dur)]
    print("%s launch duration under the profiler, ms: median %.3f over %d launches" % (kernel, sorted(big)[len(big) // 2], len(big)))
    # (the one-row launch for the empty-space constants is left out: only the full launches count)
    for k in sorted(vals):
        v = [x for x in vals[k] if x > 0.2 * max(vals[k])] if max(vals[k]) > 0 else vals[k]
        print("  " + k.ljust(34), len(v), round(sum(v) / max(len(v), 1) / 1e6, 3), "M per launch")
PY
