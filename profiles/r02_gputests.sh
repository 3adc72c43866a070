mkdir -p gpurun_out/r2
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r2/gpu_tests.log 2>&1
rc=$?
tail -25 gpurun_out/r2/gpu_tests.log
exit $rc
