# SLP vectorisation (v_pk_fma_f32 / v_pk_mul_f32) re-enabled per translation unit, fp32, against the shipped build (-fno-slp-vectorize)
mkdir -p gpurun_out/s2
L=gpurun_out/s2/slp.log; : > $L
D=$GRAFT_REPO_ROOT/quadrotor_landing_amd/csrc/build_slp
for lib in default $D/libqle_slp_tu_predict.so $D/libqle_slp_tu_step.so; do
  for B in 16384 65536 262144; do
    if [ $lib = default ]; then QLE_QUAD=0 timeout -k 10 200 python profiles/time_kernels.py $B f32 base >> $L 2>&1; else QLE_LIB=$lib QLE_QUAD=0 timeout -k 10 200 python profiles/time_kernels.py $B f32 slp >> $L 2>&1; fi
  done
done
for lib in default $D/libqle_slp_tu_quad.so; do
  if [ $lib = default ]; then QLE_QUAD=1 timeout -k 10 200 python profiles/time_kernels.py 4096 f32 base_coop >> $L 2>&1; else QLE_LIB=$lib QLE_QUAD=1 timeout -k 10 200 python profiles/time_kernels.py 4096 f32 slp_coop >> $L 2>&1; fi
done
cat $L
for lib in default $D/libqle_slp_tu_misc.so; do
  if [ $lib = default ]; then unset QLE_LIB; else export QLE_LIB=$lib; fi
  timeout -k 10 300 python bench.py --workload cfg3mr --no-cpu-baseline --no-extras --steps 1400 > gpurun_out/s2/slp_mr.json 2> gpurun_out/s2/slp_mr.err || exit 1
  python -c "
import json;d=json.load(open('gpurun_out/s2/slp_mr.json'));ms=d['ms_per_step'];print('mr lib=$lib', d['value'], ms, 'correcting tick ~', (14*ms*1e3-13*10.4), 'us')"
done
