#!/bin/bash
# usage: power_trace.sh : socket power and shader clock (rocm-smi) while k_pe_mlp16 runs back to back for ~6 s
python3 $GRAFT_REPO_ROOT/tools/micro_mlp16.py 2500 > $GRAFT_REPO_ROOT/gpurun_out/power_micro.log 2>&1 &
PID=$!
sleep 12   # workload construction + first launches
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | tr '\n' ' '; echo
  sleep 0.7
done
wait $PID
tail -1 $GRAFT_REPO_ROOT/gpurun_out/power_micro.log
echo "idle:"; sleep 2; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ' '; echo
