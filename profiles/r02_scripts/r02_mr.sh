mkdir -p gpurun_out/r2
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "multirate or replay or initialise or golden or rebase or sharding" > gpurun_out/r2/mr_tests.log 2>&1; tail -15 gpurun_out/r2/mr_tests.log
for k in 4 8 16; do
QLE_MR_K=$k timeout -k 10 300 python bench.py --workload cfg3mr --no-cpu-baseline --steps 1400 --kernel-steps 500 > gpurun_out/r2/bench_mr_k$k.json 2> gpurun_out/r2/bench_mr_k$k.err || tail -5 gpurun_out/r2/bench_mr_k$k.err
done
