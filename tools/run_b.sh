mkdir -p gpurun_out/r5d
python -m pytest tests -m gpu -q > gpurun_out/r5d/tests.log 2>&1; echo "exit $?" >> gpurun_out/r5d/tests.log; tail -5 gpurun_out/r5d/tests.log
