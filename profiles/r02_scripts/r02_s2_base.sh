# second session of round 2: whole GPU suite + bench lines with the batch-form correction as the default
mkdir -p gpurun_out/s2
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/s2/gpu_tests.log 2>&1; rc=$?
tail -3 gpurun_out/s2/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py > gpurun_out/s2/bench_cfg3.json 2> gpurun_out/s2/bench_cfg3.err && cat gpurun_out/s2/bench_cfg3.json &&
timeout -k 10 300 python bench.py --workload cfg2 > gpurun_out/s2/bench_cfg2.json 2> gpurun_out/s2/bench_cfg2.err && cat gpurun_out/s2/bench_cfg2.json
