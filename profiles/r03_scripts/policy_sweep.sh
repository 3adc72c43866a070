#!/bin/bash
# mid-size plateau: predict-only tick (HIP-event period of 300 launches) under each state cache policy / workgroup size
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3/pol; mkdir -p $O; rm -f $O/*.jsonl
for B in 131072 262144 524288; do
  for env in "" "QLE_NT=0 QLE_BLOCK=256" "QLE_NT=0 QLE_BLOCK=64" "QLE_NT=1 QLE_REFRESH=0" "QLE_NT=2 QLE_REFRESH=0" "QLE_NT=1 QLE_REFRESH=128" "QLE_NT=1 QLE_REFRESH=32" "QLE_NT=3 QLE_SPLIT=-48" "QLE_NT=3 QLE_SPLIT=-32"; do
    env $env timeout -k 10 100 python3 profiles/r03_scripts/time_predict.py $B >> $O/sweep.jsonl 2>> $O/err.log
  done
done
cat $O/sweep.jsonl
