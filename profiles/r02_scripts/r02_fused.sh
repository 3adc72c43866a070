# fused tick (QLE_STEP_MODE=2, ekf_fused.hpp) against levelled predict + batch-form correction (QLE_STEP_MODE=1), one lane per filter
mkdir -p gpurun_out/s2
L=gpurun_out/s2/fused.log
: > $L
for a in "4096 f32" "16384 f32" "65536 f32" "131072 f32" "262144 f32" "1048576 f32" "4096 f64" "16384 f64" "65536 f64" "262144 f64"; do
  for m in 1 2; do
    QLE_STEP_MODE=$m QLE_QUAD=0 timeout -k 10 200 python profiles/time_kernels.py $a mode$m >> $L 2>&1
  done
done
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/s2/fused_tests.log 2>&1; tail -3 gpurun_out/s2/fused_tests.log >> $L
cat $L
