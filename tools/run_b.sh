mkdir -p gpurun_out/r5f
python -m pytest tests -m gpu -q > gpurun_out/r5f/tests.log 2>&1; echo "exit $?" >> gpurun_out/r5f/tests.log; tail -3 gpurun_out/r5f/tests.log
