mkdir -p gpurun_out/r2
L=gpurun_out/r2/half.log
: > $L
for a in "32768 f32" "65536 f32" "98304 f32" "131072 f32" "65536 f64"; do
  for hf in 0 1; do
    QLE_HALF=$hf QLE_QUAD=0 timeout -k 10 200 python profiles/time_kernels.py $a half$hf >> $L 2>&1
  done
done
cat $L
