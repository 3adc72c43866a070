#!/bin/bash
# one GPU call for a change to the workgroup-cooperative kernel: the parity files that exercise it, the cfg 2 bench, the per-phase timelines
# usage: bash profiles/r04_scripts/kw_check.sh <outdir-name>
O=gpurun_out/r4/$1; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_population.py tests/test_gpu_compact.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?
echo tests rc=$rc; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python bench.py --workload cfg2 --no-cpu-baseline > $O/cfg2.json 2> $O/cfg2.err || exit 1
python - $O <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]+"/cfg2.json").read().strip().splitlines()[-1])
print("cfg2 ticks/s %.4g  kw_tick us %.2f  nonfinite %s  rmse %.4f"%(d["value"],d["roofline"]["avg_launch_us"],d["nonfinite_filters"],d["rmse_vs_truth"]["position_m"]))
PY
QLE_LIB=quadrotor_landing_amd/libqle_dbg.so timeout -k 10 200 python profiles/r03_scripts/kw_timeline.py 4096 f64 > $O/timeline.log 2>&1 || exit 1
cat $O/timeline.log
QLE_LIB=quadrotor_landing_amd/libqle_dbg.so timeout -k 10 200 python profiles/r03_scripts/kw_timeline.py 4096 f32 > $O/timeline32.log 2>&1 || exit 1
grep -v policy $O/timeline32.log | head -3; grep "B=4096" $O/timeline32.log | cut -c1-60
