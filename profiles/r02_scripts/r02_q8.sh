set -e
mkdir -p gpurun_out/r2
L=gpurun_out/r2/q8.log
: > $L
for a in "1024 f64" "4096 f64" "8192 f64" "16384 f64" "65536 f64"; do
    QLE_QUAD=1 QLE_ROWS_MAX=0 timeout -k 10 200 python profiles/time_kernels.py $a coop >> $L 2>&1
    QLE_QUAD=0 timeout -k 10 200 python profiles/time_kernels.py $a rows_default >> $L 2>&1
    QLE_QUAD=0 QLE_ROWS_MAX=0 timeout -k 10 200 python profiles/time_kernels.py $a lanes >> $L 2>&1
done
echo done >> $L
