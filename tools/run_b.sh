for i in 1 2; do
echo -n "new: "; python tools/micro_mlp16.py 20 2>/dev/null | tail -1
for v in e1 e2 e3; do
echo -n "$v: "; DANBO_HIP_LIB=$PWD/tools/ab/libdanbo_hip_$v.so python tools/micro_mlp16.py 20 2>/dev/null | tail -1
done
done
