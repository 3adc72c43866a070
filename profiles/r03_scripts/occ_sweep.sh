#!/bin/bash
# mid-size plateau, second experiment: the predict-only tick with the resident waves per SIMD capped by unused LDS
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3/pol; mkdir -p $O
for B in 131072 262144; do
  for env in "QLE_BLOCK=256" "QLE_BLOCK=256 QLE_LDS_PAD=65536" "QLE_BLOCK=64 QLE_LDS_PAD=40000" "QLE_BLOCK=64 QLE_LDS_PAD=20000" "QLE_BLOCK=128 QLE_LDS_PAD=65536"; do
    echo "B=$B $env: $(env $env timeout -k 10 100 python3 profiles/r03_scripts/time_predict.py $B 2>> $O/err.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["us"], "us", d["tbs"], "TB/s")')"
  done
done
