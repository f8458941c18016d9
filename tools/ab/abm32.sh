for v in "$@"; do
  if [ "$v" = base ]; then unset DANBO_HIP_LIB; else export DANBO_HIP_LIB=$PWD/tools/ab/libdanbo_hip_$v.so; fi
  echo "== $v"; timeout 300 python tools/micro_mlp32.py 10 2>&1 | grep "rel to\|k_pe_mlp16" | tail -3
done
