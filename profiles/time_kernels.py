#!/usr/bin/env python3
"""Per-kernel launch time of the hot kernels at B=65536 fp32 (HIP events, back-to-back launches)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadrotor_landing_amd as qla  # noqa: E402
from bench import CFG3  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
N = 300
out = {"batch": B, "dtype": dtype}
for name, mask_all in (("k_step_all_corrected", True), ("k_step_none_corrected", False)):
    ekf = qla.BatchedRelativePoseEKF(B, dtype, **CFG3)
    seq = ekf.make_inputs(14, np.ones(14, np.uint8))
    ekf.synth_generate(seq, seed=3)
    if not mask_all:
        for t in range(14):
            u, z, m = seq.download_tick(t)
            seq.upload_tick(t, u, z, np.zeros(B, np.uint8))
    ekf.run(seq, 0, 28); ekf.synchronize()
    ekf.timer_begin(); ekf.run(seq, 0, N); ms = ekf.timer_end()
    out[name + "_us"] = ms / N * 1e3
    ekf.close()
ekf = qla.BatchedRelativePoseEKF(B, dtype, **CFG3)
seq = ekf.make_inputs(14, None)
ekf.synth_generate(seq, seed=3)
ekf.run(seq, 0, 28); ekf.synchronize()
ekf.timer_begin(); ekf.run(seq, 0, N); ms = ekf.timer_end()
out["k_predict_us"] = ms / N * 1e3
print(json.dumps(out))
