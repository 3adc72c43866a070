#!/bin/bash
# one round of the multirate kernel work: parity tests that reach k_step_mr, per-wave timeline (diagnostic build), bench line, kernel stats
export QLE_HEAD_SHA=${1:-unknown}
TAG=${2:-mr}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3/$TAG; mkdir -p $O
timeout -k 10 500 python -m pytest tests -q -x -m gpu -k "multirate or hardware_like or recorded or twin_golden or cpp_wrapper or tick_origin" > $O/tests.log 2>&1; rc=$?
tail -3 $O/tests.log
[ $rc -ne 0 ] && exit $rc
QLE_LIB=$PWD/quadrotor_landing_amd/libqle_dbg.so timeout -k 10 200 python profiles/r03_scripts/mr_timeline.py 65536 12 > $O/mr_timeline.log 2>&1
cat $O/mr_timeline.log
timeout -k 10 300 python3 bench.py --workload cfg3mr --steps 1400 --no-cpu-baseline > $O/bench_mr.json 2> $O/bench_mr.err || tail -5 $O/bench_mr.err
python3 -c "
import json; d=json.load(open('$O/bench_mr.json')); print('cfg3mr ticks/s %.3e  us/step %.2f  predict us %.2f' % (d['value'], d['ms_per_step']*1e3, d['roofline']['avg_launch_us']))"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_mr -o s -- python3 bench.py --workload cfg3mr --no-cpu-baseline --no-extras --steps 1400 > $O/stats_mr.log 2>&1
python3 profiles/summarize.py $O/stats_mr $O/kernel_stats_multirate.md "bench.py --workload cfg3mr --steps 1400"
sed -n 7,12p $O/kernel_stats_multirate.md | cut -c1-160
python3 profiles/r03_scripts/after_step.py $O/stats_mr k_step_mr > $O/after_step.md; cat $O/after_step.md
