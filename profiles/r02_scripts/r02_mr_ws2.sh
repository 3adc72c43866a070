# multirate correcting tick: IMU-sample prefetch in both kernels; wave-specialised (QLE_MR_WS=1) vs one lane (0); kernel times by rocprofv3
mkdir -p gpurun_out/s2
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "multirate or recorded or stamped or mr" > gpurun_out/s2/ws_tests.log 2>&1; rc=$?
tail -3 gpurun_out/s2/ws_tests.log
[ $rc -eq 0 ] || exit $rc
R=$GRAFT_REPO_ROOT/gpurun_out/s2
for ws in 0 1; do
QLE_MR_WS=$ws timeout -k 10 300 python bench.py --workload cfg3mr --no-cpu-baseline --no-extras > $R/bench_mr_ws$ws.json 2> $R/bench_mr_ws$ws.err || exit 1
python -c "
import json;d=json.load(open('gpurun_out/s2/bench_mr_ws$ws.json'));print('ws=$ws',d['value'],d['ms_per_step'],d['nonfinite_filters'])"
done
export TMPDIR=/tmp
for ws in 0 1; do
export QLE_MR_WS=$ws
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/stats_mr_ws$ws -o s -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg3mr --no-cpu-baseline --no-extras --steps 1400 > $R/stats_mr_ws$ws.log 2>&1) || exit 1
python profiles/summarize.py $R/stats_mr_ws$ws $R/stats_mr_ws$ws.md ws$ws > /dev/null && grep -E "k_step_mr|k_predict" $R/stats_mr_ws$ws.md | cut -c1-200
done
