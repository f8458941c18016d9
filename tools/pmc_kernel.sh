#!/bin/bash
# usage: pmc_kernel.sh TAG KERNEL_SUBSTRING [bench_train args] : SQ counter passes over 8 training steps, averaged per launch of one kernel
#        PMC_FRAME=1 pmc_kernel.sh TAG KERNEL_SUBSTRING : the same over three render frames of bench.py (config 1)
TAG=$1; KN=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmck_$TAG
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_INSTS_FLAT" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  if [ -n "$PMC_FRAME" ]; then
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dense --no-sweep > $OUT.p$i.log 2>&1
  else
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- python3 $GRAFT_REPO_ROOT/tools/bench_train.py --steps 5 --warmup 3 "$@" > $OUT.p$i.log 2>&1
  fi
done
KN=$KN python3 - <<PY
import csv, glob, os, collections
tot = collections.defaultdict(float); calls = collections.Counter()
for f in glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmck_$TAG/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if os.environ["KN"] in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); calls[r["Counter_Name"]] += 1
for k in sorted(tot):
    print(k.ljust(34), calls[k], round(tot[k] / calls[k] / 1e6, 4), "M per launch")
PY
