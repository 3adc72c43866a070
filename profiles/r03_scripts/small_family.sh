#!/bin/bash
# small batches: lane-per-filter kernels with one-wave workgroups against the workgroup-cooperative kernel (all ticks / ticks with tag poses)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in "1024 f64" "2048 f64" "4096 f64" "8192 f64" "16384 f64" "1024 f32" "4096 f32" "8192 f32" "16384 f32"; do
  for env in "QLE_QUAD=0 QLE_BLOCK=64" "QLE_QUAD=3" "QLE_QUAD=1 QLE_BLOCK=64"; do
    set -- $spec
    echo "$spec [$env]: $(env $env timeout -k 10 100 python3 profiles/time_kernels.py $1 $2 x 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("step_all", d["step_all_us"], "step_none", d["step_none_us"], "predict", d["predict_us"])')"
  done
done
