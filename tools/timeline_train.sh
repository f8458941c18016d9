#!/bin/bash
# usage: timeline_train.sh TAG [bench_train args] -> start/end of every kernel of the last captured training step, relative to the
# step's first kernel (rocprofv3 kernel trace); shows what runs beside what and where the streams idle
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl_$TAG -o $TAG -- python3 $GRAFT_REPO_ROOT/tools/bench_train.py --steps 20 --warmup 5 "$@" > $GRAFT_REPO_ROOT/gpurun_out/tl_${TAG}.log 2>&1
python3 - <<PY
import csv,os,glob
f=glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/tl_$TAG/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# a step begins at the zeroing memset / first kernel after k_adam
idx=[i for i,r in enumerate(rows) if "k_adam" in r["Kernel_Name"]]
a,b=idx[-3]+1,idx[-2]+1
t0=int(rows[a]["Start_Timestamp"])
out=[]
for r in rows[a:b]:
    s=(int(r["Start_Timestamp"])-t0)/1e3; e=(int(r["End_Timestamp"])-t0)/1e3
    out.append("%8.1f %8.1f %7.1f  q%-3s %s"%(s,e,e-s,r.get("Queue_Id","?"),r["Kernel_Name"][:60]))
open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/tl_$TAG.txt","w").write("\n".join(out)+"\n")
print("\n".join(out))
PY
