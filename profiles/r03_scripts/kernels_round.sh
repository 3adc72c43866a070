#!/bin/bash
# per-kernel launch times of the single-rate hot kernels (HIP-event period of 300 back-to-back launches) over the sizes the
# review names, one-lane kernels (QLE_QUAD=0) and the default policy at the small fp64 batch
cd $GRAFT_REPO_ROOT; TAG=${1:-k}; O=gpurun_out/r3/$TAG; mkdir -p $O
for spec in "16384 f32" "65536 f32" "131072 f32" "262144 f32" "1048576 f32" "65536 f64" "4096 f64"; do
  set -- $spec
  QLE_QUAD=0 timeout -k 10 120 python3 profiles/time_kernels.py $1 $2 lanes >> $O/times.jsonl 2>> $O/err.log
done
timeout -k 10 120 python3 profiles/time_kernels.py 4096 f64 default >> $O/times.jsonl 2>> $O/err.log
timeout -k 10 120 python3 profiles/time_kernels.py 4096 f32 default >> $O/times.jsonl 2>> $O/err.log
python3 - <<PY
import json
for l in open("$O/times.jsonl"):
    d = json.loads(l); print("%8d %s %-8s step_all %7.2f  step_none %7.2f  predict %7.2f" % (d["batch"], d["dtype"], d["label"], d["step_all_us"], d["step_none_us"], d["predict_us"]))
PY
