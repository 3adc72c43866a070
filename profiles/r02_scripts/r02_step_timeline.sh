mkdir -p gpurun_out/s2
for B in 65536 16384 131072; do
QLE_QUAD=0 QLE_LIB=$GRAFT_REPO_ROOT/quadrotor_landing_amd/csrc/build/libqle_dbg.so timeout -k 10 200 python profiles/r02_scripts/r02_step_timeline.py $B
done 2>&1 | tee gpurun_out/s2/step_timeline.log
