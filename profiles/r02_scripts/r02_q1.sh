set -e
mkdir -p gpurun_out/r2
L=gpurun_out/r2/q1.log
: > $L
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "quad-forced" > gpurun_out/r2/q1_pytest.log 2>&1 || { tail -30 gpurun_out/r2/q1_pytest.log; echo "PYTEST FAILED" >> $L; }
tail -3 gpurun_out/r2/q1_pytest.log >> $L
for a in "65536 f32" "262144 f32" "65536 f64" "4096 f64" "4096 f32"; do
  for q in 0 1 3; do
    QLE_QUAD=$q QLE_ROWS_MAX=0 timeout -k 10 200 python profiles/time_kernels.py $a >> $L 2>&1
  done
done
echo done >> $L
