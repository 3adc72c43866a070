mkdir -p gpurun_out/r2
L=gpurun_out/r2/q9.log
: > $L
for a in "65536 f32" "131072 f32" "262144 f32" "65536 f64"; do
    timeout -k 10 200 python profiles/time_kernels.py $a >> $L 2>&1
done
timeout -k 10 600 python bench.py --no-cpu-baseline > gpurun_out/r2/bench_cfg3_b.json 2> gpurun_out/r2/bench_cfg3_b.err || tail -20 gpurun_out/r2/bench_cfg3_b.err
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r2/gpu_tests2.log 2>&1; tail -3 gpurun_out/r2/gpu_tests2.log >> $L
cat $L
