#!/bin/bash
# NOTE: the environment switches / variant libraries this script drives belonged to an experiment build that is not in the tree
# (what was changed is described in profiles/r03_tuning.md section 3; its log is under profiles/r03_logs/).
# explicit scope bits on the stores of the multirate history (experiment builds libqle_v<N>.so: make OBJDIR=build_vN OUT=../libqle_vN.so
# EXTRA=-DQLE_HIST_SCOPE=N; 1 "sc0 sc1 nt", 2 "sc1", 3 "sc0 sc1", 4 "nt sc1"): bench line, kernel stats, ticks after the correcting tick
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3/${1:-hs}; mkdir -p $O
for v in 0 1 2 3 4 5 6 7; do
  lib=$PWD/quadrotor_landing_amd/libqle_v$v.so; [ $v = 0 ] && lib=$PWD/quadrotor_landing_amd/libqle_ekf.so
  [ -f $lib ] || continue
  export QLE_LIB=$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/v$v -o s -- python3 bench.py --workload cfg3mr --no-cpu-baseline --no-extras --steps 1400 > $O/v$v.json 2> $O/v$v.err
  echo "== variant $v: $(python3 -c "
import json; d=json.load(open('$O/v$v.json')); print('ticks/s %.3e us/step %.3f nonfinite %d' % (d['value'], d['ms_per_step']*1e3, d['nonfinite_filters']))")"
  python3 profiles/r03_scripts/after_step.py $O/v$v k_step_mr | sed -n 3,10p
done 2>&1 | tee $O/summary.md
