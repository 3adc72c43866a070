#!/bin/bash
# small states: is the cached-store refresh tick needed at all below the 36 MiB state it was tuned on?  sustained rates (0.6 s each)
cd $GRAFT_REPO_ROOT
for B in 16384 32768 49152; do
  for env in "" "QLE_NT=2 QLE_REFRESH=0" "QLE_NT=1 QLE_REFRESH=1024"; do
    echo "B=$B [$env]: $(env $env timeout -k 10 100 python3 profiles/time_sustained.py $B predict 2>/dev/null | tail -1)"
  done
done
