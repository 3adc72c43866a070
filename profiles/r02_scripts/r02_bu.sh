mkdir -p gpurun_out/r2
L=gpurun_out/r2/bu.log
: > $L
for a in "16384 f32" "32768 f32" "65536 f32" "131072 f32" "262144 f32" "1048576 f32"; do
  for bu in 0 1; do
    QLE_STEP_BATCH=$bu QLE_QUAD=0 timeout -k 10 200 python profiles/time_kernels.py $a bu$bu >> $L 2>&1
  done
done
QLE_STEP_BATCH=1 timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "lanes-only" > gpurun_out/r2/bu_tests.log 2>&1; tail -3 gpurun_out/r2/bu_tests.log >> $L
cat $L
