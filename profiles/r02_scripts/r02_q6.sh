set -e
mkdir -p gpurun_out/r2
L=gpurun_out/r2/q6.log
: > $L
for lib in "" _ilp _dflt; do
for a in "65536 f32" "262144 f32" "65536 f64"; do
    QLE_LIB=$PWD/quadrotor_landing_amd/libqle_ekf$lib.so QLE_QUAD=0 QLE_ROWS_MAX=0 timeout -k 10 200 python profiles/time_kernels.py $a >> $L 2>&1
done
done
echo done >> $L
