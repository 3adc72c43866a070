#!/bin/bash
# NOTE: the environment switches / variant libraries this script drives belonged to an experiment build that is not in the tree
# (what was changed is described in profiles/r03_tuning.md section 3; its log is under profiles/r03_logs/).
# what the ticks after a correcting multirate tick cost, by cache policy of the extra checkpoint (QLE_CK_CACHED) and of the state accesses of
# k_step_mr (QLE_MR_NT): kernel stats + duration by distance from the correcting tick
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3/${1:-mra}; mkdir -p $O
for ck in 1 0; do for nt in 1 0; do
  tag=ck${ck}_nt${nt}
  export QLE_CK_CACHED=$ck QLE_MR_NT=$nt
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -o s -- python3 bench.py --workload cfg3mr --no-cpu-baseline --no-extras --steps 1400 > $O/$tag.json 2> $O/$tag.err
  echo "== QLE_CK_CACHED=$ck QLE_MR_NT=$nt: $(python3 -c "
import json; d=json.load(open('$O/$tag.json')); print('ticks/s %.3e us/step %.3f' % (d['value'], d['ms_per_step']*1e3))")"
  python3 profiles/r03_scripts/after_step.py $O/$tag k_step_mr | sed -n 3,11p
done; done 2>&1 | tee $O/summary.md
