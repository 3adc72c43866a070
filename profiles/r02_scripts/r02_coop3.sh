mkdir -p gpurun_out/r2
L=gpurun_out/r2/coop3.log
: > $L
for a in "1024 f32" "8192 f32" "16384 f32" "32768 f32" "49152 f32" "24576 f64"; do
    QLE_QUAD=1 timeout -k 10 200 python profiles/time_kernels.py $a coop >> $L 2>&1
    QLE_QUAD=0 timeout -k 10 200 python profiles/time_kernels.py $a lanes >> $L 2>&1
done
cat $L
