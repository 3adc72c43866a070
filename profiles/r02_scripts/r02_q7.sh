set -e
mkdir -p gpurun_out/r2
L=gpurun_out/r2/q7.log
: > $L
for a in "65536 f32" "131072 f32" "262144 f32" "1048576 f32" "65536 f64" "4096 f64" "4096 f32"; do
    QLE_QUAD=0 QLE_ROWS_MAX=0 timeout -k 10 200 python profiles/time_kernels.py $a >> $L 2>&1
done
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lanes-only or default" > gpurun_out/r2/q7_pytest.log 2>&1 || { tail -30 gpurun_out/r2/q7_pytest.log; echo "PYTEST FAILED" >> $L; }
tail -3 gpurun_out/r2/q7_pytest.log >> $L
echo done >> $L
