# wave-specialised multirate correcting tick, IMU samples in static register slots loaded 4 (default build) / 2 phases ahead
mkdir -p gpurun_out/s2
R=$GRAFT_REPO_ROOT/gpurun_out/s2
QLE_MR_WS=1 timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "multirate or recorded or stamped or mr" 2>&1 | tail -2
run() {  # label lib ws
  QLE_LIB=$2 QLE_MR_WS=$3 timeout -k 10 300 python bench.py --workload cfg3mr --no-cpu-baseline --no-extras --steps 1400 > $R/bench_ws4_$1.json 2> $R/bench_ws4.err || exit 1
  python -c "
import json;d=json.load(open('$R/bench_ws4_$1.json'));ms=d['ms_per_step'];print('$1', d['value'], ms, 'correcting tick ~', (14*ms*1e3-13*10.4), 'us')"
}
run lane $GRAFT_REPO_ROOT/quadrotor_landing_amd/libqle_ekf.so 0
run ws_a4 $GRAFT_REPO_ROOT/quadrotor_landing_amd/libqle_ekf.so 1
run ws_a2 $GRAFT_REPO_ROOT/quadrotor_landing_amd/csrc/build_ws/libqle_ws_a2.so 1
export TMPDIR=/tmp QLE_MR_WS=1
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/stats_ws4 -o s -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg3mr --no-cpu-baseline --no-extras --steps 1400 > $R/stats_ws4.log 2>&1)
python profiles/summarize.py $R/stats_ws4 $R/stats_ws4.md ws4 > /dev/null && grep -E "k_step_mr" $R/stats_ws4.md | cut -c1-160
