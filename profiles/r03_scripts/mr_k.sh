#!/bin/bash
# checkpoint period of the multirate history with the extra (expected-entry) checkpoint in place: cfg3mr bench line per QLE_MR_K
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3/mrk; mkdir -p $O
for K in 8 16 32 64; do
  QLE_MR_K=$K timeout -k 10 200 python3 bench.py --workload cfg3mr --steps 1400 --no-cpu-baseline --no-extras > $O/k$K.json 2> $O/k$K.err
  python3 -c "
import json; d=json.load(open('$O/k$K.json')); print('k=$K cfg3mr ticks/s %.3e  us/step %.2f  predict tick us %.2f' % (d['value'], d['ms_per_step']*1e3, d['roofline']['avg_launch_us']))"
done
