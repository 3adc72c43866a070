# wave-specialised multirate correcting tick with the IMU samples loaded 4 / 8 steps ahead, against the one-lane kernel (QLE_MR_WS=0)
mkdir -p gpurun_out/s2
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k 'fused_tick_lanes' 2>&1 | tail -3
for a in 4 8; do
export QLE_LIB=$GRAFT_REPO_ROOT/quadrotor_landing_amd/csrc/build_ws/libqle_ws_a$a.so
QLE_MR_WS=1 timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "multirate or recorded or stamped or mr" 2>&1 | tail -2
for ws in 0 1; do
QLE_MR_WS=$ws timeout -k 10 300 python bench.py --workload cfg3mr --no-cpu-baseline --no-extras --steps 1400 > gpurun_out/s2/bench_ws3_a${a}_$ws.json 2> gpurun_out/s2/bench_ws3.err || exit 1
python -c "
import json;d=json.load(open('gpurun_out/s2/bench_ws3_a${a}_$ws.json'));ms=d['ms_per_step'];print('ahead=$a ws=$ws', d['value'], ms, 'correcting tick ~', (14*ms*1e3-13*10.5), 'us')"
done
done
