#!/bin/bash
# small states (L2-sized): state cache policy of the ticks
cd $GRAFT_REPO_ROOT
for spec in "1024 f64" "4096 f64" "4096 f32" "16384 f64" "16384 f32" "32768 f32"; do
  for env in "QLE_NT=1" "QLE_NT=0" "QLE_NT=2 QLE_REFRESH=0"; do
    set -- $spec
    echo "$spec [$env]: $(env $env timeout -k 10 100 python3 profiles/time_kernels.py $1 $2 x 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("step_all", d["step_all_us"], "step_none", d["step_none_us"], "predict", d["predict_us"])')"
  done
done
