mkdir -p gpurun_out/r2
timeout -k 10 600 python bench.py > gpurun_out/r2/bench_cfg3.json 2> gpurun_out/r2/bench_cfg3.err || tail -20 gpurun_out/r2/bench_cfg3.err
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2/bench_cfg3_driver.json 2> gpurun_out/r2/bench_cfg3_driver.err || tail -20 gpurun_out/r2/bench_cfg3_driver.err
timeout -k 10 300 python bench.py --workload cfg2 > gpurun_out/r2/bench_cfg2.json 2> gpurun_out/r2/bench_cfg2.err || tail -20 gpurun_out/r2/bench_cfg2.err
timeout -k 10 300 python bench.py --workload cfg5 --steps 1400 > gpurun_out/r2/bench_cfg5.json 2> gpurun_out/r2/bench_cfg5.err || tail -20 gpurun_out/r2/bench_cfg5.err
timeout -k 10 300 python bench.py --workload cfg4 --steps 1400 > gpurun_out/r2/bench_cfg4.json 2> gpurun_out/r2/bench_cfg4.err || tail -20 gpurun_out/r2/bench_cfg4.err
timeout -k 10 600 python -m pytest tests/test_bench_contract.py -x -q -m gpu > gpurun_out/r2/bench_tests.log 2>&1; tail -5 gpurun_out/r2/bench_tests.log
