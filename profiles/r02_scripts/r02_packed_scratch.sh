# scratch build (not in the tree: pairing covariance order, fp32 downdate / V / NV / dx with v_pk_fma_f32; profiles/micro/packed_order) against the shipped build
mkdir -p gpurun_out/s2
L=gpurun_out/s2/packed_scratch.log; : > $L
PK=$GRAFT_REPO_ROOT/quadrotor_landing_amd/csrc/build/libqle_pk.so
QLE_QUAD=0 QLE_LIB=$PK timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lanes-only and ((fused_step_equals and f32) or (predict_teacher_forced and f32) or (golden_sequences))" 2>&1 | tail -3 >> $L
for B in 16384 65536 131072 262144; do
  QLE_QUAD=0 timeout -k 10 200 python profiles/time_kernels.py $B f32 shipped >> $L 2>&1
  QLE_QUAD=0 QLE_LIB=$PK timeout -k 10 200 python profiles/time_kernels.py $B f32 packed_scratch >> $L 2>&1
done
cat $L
