mkdir -p gpurun_out/r5b
python -m pytest tests -m gpu -q > gpurun_out/r5b/tests.log 2>&1; echo "exit $?" >> gpurun_out/r5b/tests.log; tail -8 gpurun_out/r5b/tests.log
for i in 1 2; do
python tools/micro_mlp16.py 20 2>/dev/null | tail -1
DANBO_MLP16_NOSCALE=1 python tools/micro_mlp16.py 20 2>/dev/null | tail -1
done
