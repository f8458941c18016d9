mkdir -p gpurun_out/r5g
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_trunk.py tests/test_gpu_train_engine.py tests/test_gpu_fullsize.py -q -x > gpurun_out/r5g/tests.log 2>&1; echo "exit $?" >> gpurun_out/r5g/tests.log; tail -3 gpurun_out/r5g/tests.log
for i in 1 2; do
for v in base tail; do
echo -n "$v: "; DANBO_HIP_LIB=$PWD/tools/ab/libdanbo_hip_$v.so python tools/micro_mlp16.py 20 2>/dev/null | tail -1
echo -n "$v train: "; DANBO_HIP_LIB=$PWD/tools/ab/libdanbo_hip_$v.so python tools/bench_train.py 2>/dev/null | tail -1
done
done
