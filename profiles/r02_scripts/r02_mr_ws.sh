# wave-specialised multirate correcting tick (k_step_mr_ws) against the one-lane kernel (QLE_MR_WS=0)
mkdir -p gpurun_out/s2
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "multirate or recorded or stamped or mr" > gpurun_out/s2/ws_tests.log 2>&1; rc=$?
tail -3 gpurun_out/s2/ws_tests.log
[ $rc -eq 0 ] || exit $rc
for ws in 0 1; do
QLE_MR_WS=$ws timeout -k 10 300 python bench.py --workload cfg3mr --no-cpu-baseline --no-extras > gpurun_out/s2/bench_mr_ws$ws.json 2> gpurun_out/s2/bench_mr_ws$ws.err || exit 1
python -c "
import json;d=json.load(open('gpurun_out/s2/bench_mr_ws$ws.json'));print('ws=$ws',d['value'],d['ms_per_step'],d['nonfinite_filters'],d['rmse_vs_truth'])"
done
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/s2/ws_tests_all.log 2>&1; tail -3 gpurun_out/s2/ws_tests_all.log
