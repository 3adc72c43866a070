mkdir -p gpurun_out/r2
L=gpurun_out/r2/coop2.log
: > $L
for a in "1024 f64" "4096 f64" "16384 f64" "32768 f64" "4096 f32"; do
    QLE_QUAD=1 timeout -k 10 200 python profiles/time_kernels.py $a coop >> $L 2>&1
    QLE_QUAD=0 timeout -k 10 200 python profiles/time_kernels.py $a lanes >> $L 2>&1
done
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "coop-forced or default" > gpurun_out/r2/coop2_tests.log 2>&1; tail -3 gpurun_out/r2/coop2_tests.log >> $L
timeout -k 10 300 python bench.py --workload cfg2 >> $L 2>/dev/null
cat $L
