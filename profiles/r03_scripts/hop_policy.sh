#!/bin/bash
# NOTE: the environment switches / variant libraries this script drives belonged to an experiment build that is not in the tree
# (what was changed is described in profiles/r03_tuning.md section 3; its log is under profiles/r03_logs/).
# zero-copy checkpoints: cache policy of the tick that writes the state into a checkpoint slot (QLE_HOP_CACHED: cached stores) and of the
# tick that brings it home (QLE_HOP_BACK: cached stores)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3/${1:-hop}; mkdir -p $O
for c in "0 0" "0 1" "1 1" "1 0"; do
  set -- $c; export QLE_HOP_CACHED=$1 QLE_HOP_BACK=$2; tag=hop$1$2
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -o s -- python3 bench.py --workload cfg3mr --no-cpu-baseline --no-extras --steps 1400 > $O/$tag.json 2> $O/$tag.err
  echo "== QLE_HOP_CACHED=$1 QLE_HOP_BACK=$2: $(python3 -c "
import json; d=json.load(open('$O/$tag.json')); print('ticks/s %.3e us/step %.3f' % (d['value'], d['ms_per_step']*1e3))")"
  python3 profiles/r03_scripts/after_step.py $O/$tag k_step_mr | sed -n 3,11p
done 2>&1 | tee $O/summary.md
