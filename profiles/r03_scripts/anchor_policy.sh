#!/bin/bash
# NOTE: the environment switches / variant libraries this script drives belonged to an experiment build that is not in the tree
# (what was changed is described in profiles/r03_tuning.md section 3; its log is under profiles/r03_logs/).
# cache policy of the anchor stores of k_step_mr (QLE_ANCHOR_CACHED=1: cached stores, so that the anchor slot stays in the Infinity Cache
# next to the state and the extra checkpoint and its rewrite every cycle never reaches HBM)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3/${1:-anc}; mkdir -p $O
for c in 0 1; do
  export QLE_ANCHOR_CACHED=$c; tag=anc$c
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -o s -- python3 bench.py --workload cfg3mr --no-cpu-baseline --no-extras --steps 1400 > $O/$tag.json 2> $O/$tag.err
  echo "== QLE_ANCHOR_CACHED=$c: $(python3 -c "
import json; d=json.load(open('$O/$tag.json')); print('ticks/s %.3e us/step %.3f' % (d['value'], d['ms_per_step']*1e3))")"
  python3 profiles/r03_scripts/after_step.py $O/$tag k_step_mr | sed -n 3,11p
done 2>&1 | tee $O/summary.md
