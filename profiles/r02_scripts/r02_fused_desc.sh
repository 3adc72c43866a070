# fused tick with bottom-up loads and the rows r (level 3) last, against the ascending order (build with -DQLE_FUSED_DESC=0 kept as QLE_LIB if present)
mkdir -p gpurun_out/s2
L=gpurun_out/s2/fused_desc.log; : > $L
for a in "16384 f32" "65536 f32" "131072 f32" "262144 f32" "1048576 f32" "4096 f64" "65536 f64" "262144 f64"; do
  QLE_QUAD=0 timeout -k 10 200 python profiles/time_kernels.py $a desc >> $L 2>&1
done
cat $L
timeout -k 10 800 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
