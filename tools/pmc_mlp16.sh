#!/bin/bash
# usage: pmc_mlp16.sh TAG : SQ counter passes over the MLP16 micro-benchmark (each --pmc set in its own run)
TAG=$1
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- python3 $GRAFT_REPO_ROOT/tools/micro_mlp16.py 3 > $OUT.p$i.log 2>&1
  tail -1 $OUT.p$i.log
done
python3 - <<PY
import csv, glob, os, collections
tot = collections.defaultdict(float); calls = collections.Counter()
for f in glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_$TAG/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_pe_mlp16" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); calls[r["Counter_Name"]] += 1
for k in sorted(tot):
    print(k.ljust(34), calls[k], round(tot[k] / calls[k] / 1e6, 3), "M per launch")
PY
