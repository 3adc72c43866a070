#!/bin/bash
# below 65 536 filters a 256-thread workgroup leaves CUs idle (32 768 filters = 128 workgroups on 256 CUs): workgroup size of the
# lane-per-filter kernels at 8 192 ... 49 152 filters
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in "8192 f32" "16384 f32" "32768 f32" "49152 f32" "16384 f64" "32768 f64"; do
  for blk in 256 128 64; do
    set -- $spec
    echo "$spec block $blk: $(QLE_QUAD=0 QLE_BLOCK=$blk timeout -k 10 100 python3 profiles/time_kernels.py $1 $2 x 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("step_all", d["step_all_us"], "step_none", d["step_none_us"], "predict", d["predict_us"])')"
  done
done
