# fused tick: innovation, its elimination, dx and the injection after the sweep (fake dependency on the last covariance word)
mkdir -p gpurun_out/s2
L=gpurun_out/s2/innov_late.log; : > $L
for a in "4096 f32" "16384 f32" "65536 f32" "131072 f32" "262144 f32" "4096 f64" "16384 f64" "65536 f64" "262144 f64"; do
  QLE_QUAD=0 timeout -k 10 200 python profiles/time_kernels.py $a innov_late >> $L 2>&1
done
cat $L
timeout -k 10 800 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
