mkdir -p gpurun_out/r5h
python -m pytest tests -m gpu -q > gpurun_out/r5h/tests.log 2>&1; echo "exit $?" >> gpurun_out/r5h/tests.log; tail -4 gpurun_out/r5h/tests.log
for i in 1 2 3; do python tools/bench_train.py 2>/dev/null | tail -1 | cut -c1-120; done
