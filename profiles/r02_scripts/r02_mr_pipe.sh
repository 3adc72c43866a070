# software-pipelined multirate replay: whole GPU suite, cfg3mr bench line, per-kernel times under rocprofv3
mkdir -p gpurun_out/s2
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/s2/mr_tests.log 2>&1; rc=$?
tail -3 gpurun_out/s2/mr_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --workload cfg3mr --no-cpu-baseline --no-extras > gpurun_out/s2/bench_mr.json 2> gpurun_out/s2/bench_mr.err && cat gpurun_out/s2/bench_mr.json &&
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/s2/prof_mr -o mr -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg3mr --no-cpu-baseline --no-extras --steps 1400 > $GRAFT_REPO_ROOT/gpurun_out/s2/prof_mr.log 2>&1
cd $GRAFT_REPO_ROOT && python profiles/summarize.py gpurun_out/s2/prof_mr gpurun_out/s2/prof_mr.md mr && cat gpurun_out/s2/prof_mr.md
