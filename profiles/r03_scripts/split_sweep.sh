#!/bin/bash
# large states: cached share (k of every 64 workgroup groups) of the split policy, predict-only tick
cd $GRAFT_REPO_ROOT
for B in 1048576 2097152 4194304; do
  for env in "" "QLE_NT=3 QLE_SPLIT=-2" "QLE_NT=3 QLE_SPLIT=-4" "QLE_NT=3 QLE_SPLIT=-6" "QLE_NT=3 QLE_SPLIT=-8" "QLE_NT=3 QLE_SPLIT=-12" "QLE_NT=3 QLE_SPLIT=-16" "QLE_NT=2 QLE_REFRESH=0" "QLE_NT=0"; do
    echo "B=$B [$env]: $(env $env timeout -k 10 150 python3 profiles/r03_scripts/time_predict.py $B 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["us"], "us", d["tbs"], "TB/s policy", d["policy"])')"
  done
done
