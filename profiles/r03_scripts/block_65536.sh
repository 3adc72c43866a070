#!/bin/bash
# workgroup size of the lane-per-filter kernels at 65 536 and 131 072 filters (level: the 256-thread rule stays there; below 65 536 see small_block.sh)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in "65536 f32" "131072 f32" "65536 f64"; do
  for blk in 256 128 64; do
    set -- $spec
    echo "$spec block $blk: $(QLE_QUAD=0 QLE_BLOCK=$blk timeout -k 10 100 python3 profiles/time_kernels.py $1 $2 x 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("step_all", d["step_all_us"], "step_none", d["step_none_us"], "predict", d["predict_us"])')"
  done
done
