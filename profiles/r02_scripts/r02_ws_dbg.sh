# where the wave-specialised correcting tick spends its time: variants with the nominal part (1), the covariance part (2) or both (3) removed
mkdir -p gpurun_out/s2
for v in 1 2 3; do
QLE_LIB=$GRAFT_REPO_ROOT/quadrotor_landing_amd/csrc/build/libqle_dbg$v.so QLE_MR_WS=1 timeout -k 10 300 python bench.py --workload cfg3mr --no-cpu-baseline --no-extras --steps 1400 > gpurun_out/s2/bench_ws_dbg$v.json 2> gpurun_out/s2/bench_ws_dbg$v.err || exit 1
python -c "
import json;d=json.load(open('gpurun_out/s2/bench_ws_dbg$v.json'));ms=d['ms_per_step'];print('dbg$v', ms, 'correcting tick ~', (14*ms*1e3-13*10.5), 'us')"
done
