mkdir -p gpurun_out/r2
L=gpurun_out/r2/bu_small.log
: > $L
for a in "1024 f64" "4096 f64" "8192 f64" "16384 f64" "24576 f64" "1024 f32" "4096 f32" "8192 f32" "16384 f32"; do
    QLE_STEP_BATCH=1 QLE_QUAD=0 timeout -k 10 200 python profiles/time_kernels.py $a lanes_bu >> $L 2>&1
    QLE_STEP_BATCH=1 QLE_QUAD=1 timeout -k 10 200 python profiles/time_kernels.py $a coop >> $L 2>&1
done
QLE_STEP_BATCH=1 timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r2/bu_tests2.log 2>&1; tail -3 gpurun_out/r2/bu_tests2.log >> $L
cat $L
