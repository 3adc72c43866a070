mkdir -p gpurun_out/r2
L=gpurun_out/r2/bu64.log
: > $L
for a in "32768 f64" "65536 f64" "262144 f64"; do
  for bu in 0 1; do
    QLE_STEP_BATCH=$bu QLE_QUAD=0 timeout -k 10 200 python profiles/time_kernels.py $a bu$bu >> $L 2>&1
  done
done
cat $L
