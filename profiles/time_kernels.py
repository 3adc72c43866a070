#!/usr/bin/env python3
"""Per-kernel launch time of the hot kernels (HIP events, back-to-back launches): a tick whose tag record corrects every
filter, one whose mask is all zero, and a predict-only tick.

    [QLE_QUAD=0|1|2|3|7] [QLE_LIB=...] [QLE_TIME_DIRECT=0] python profiles/time_kernels.py <batch> <f32|f64> [label]

QLE_TIME_DIRECT=0 times the conventional orientation method (direct_orien_method = 0, the reference's default, EKF.cpp:440-444);
QLE_TIME_EST_BIAS=0 the 9-state filter (est_bias = false, EKF.cpp:92; compact records wherever the lane-per-filter kernels serve every tick, i.e. above 4 096 filters; QLE_COMPACT=0|1 forces).
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadrotor_landing_amd as qla  # noqa: E402
from bench import CFG3  # noqa: E402

CFG3 = dict(CFG3, direct_orien_method=int(os.environ.get("QLE_TIME_DIRECT", "1")), est_bias=int(os.environ.get("QLE_TIME_EST_BIAS", "1")))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
N = int(os.environ.get("QLE_TIME_N", "300"))
out = {"label": sys.argv[3] if len(sys.argv) > 3 else "", "batch": B, "dtype": dtype, "direct": CFG3["direct_orien_method"], "est_bias": CFG3["est_bias"], "compact": os.environ.get("QLE_COMPACT", "auto"), "quad": os.environ.get("QLE_QUAD", "auto"),
       "lib": os.path.basename(os.environ.get("QLE_LIB", "default"))}
for name, mask_all in (("step_all_us", True), ("step_none_us", False)):
    ekf = qla.BatchedRelativePoseEKF(B, dtype, **CFG3)
    seq = ekf.make_inputs(14, np.ones(14, np.uint8))
    ekf.synth_generate(seq, seed=3)
    if not mask_all:
        for t in range(14):
            u, z, m = seq.download_tick(t)
            seq.upload_tick(t, u, z, np.zeros(B, np.uint8))
    ekf.run(seq, 0, 28); ekf.synchronize()
    ekf.timer_begin(); ekf.run(seq, 0, N); ms = ekf.timer_end()
    out[name] = round(ms / N * 1e3, 2)
    out["bad_" + name[:-3]] = ekf.count_nonfinite()
    ekf.close()
ekf = qla.BatchedRelativePoseEKF(B, dtype, **CFG3)
seq = ekf.make_inputs(14, None)
ekf.synth_generate(seq, seed=3)
ekf.run(seq, 0, 28); ekf.synchronize()
ekf.timer_begin(); ekf.run(seq, 0, N); ms = ekf.timer_end()
out["predict_us"] = round(ms / N * 1e3, 2)
out["record_words"] = ekf.policy()["record_words"]
out["predict_bytes"] = ekf.algorithmic_bytes(0)
out["predict_GBs"] = round(ekf.algorithmic_bytes(0) / (ms / N * 1e-3) / 1e9, 1)
print(json.dumps(out), flush=True)
