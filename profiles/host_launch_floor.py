#!/usr/bin/env python3
"""Host launch floor of qle_run: with a tiny batch the kernels are shorter than the host-side launch
path, so wall time per tick is the rate at which one rank can issue ticks."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadrotor_landing_amd as qla
from bench import CFG3
out = {}
for B in (256, 65536):
    T = 1400
    thm = np.zeros(T, np.uint8); thm[13::14] = 1
    ekf = qla.BatchedRelativePoseEKF(B, "f32", **CFG3)
    seq = ekf.make_inputs(T, thm)
    ekf.synth_generate(seq, seed=1)
    ekf.run(seq, 0, 140); ekf.synchronize()
    t0 = time.perf_counter(); ekf.run(seq, 0, 4 * T); t_issue = time.perf_counter() - t0
    ekf.synchronize(); t_all = time.perf_counter() - t0
    out[B] = {"us_per_tick_issue_only": t_issue / (4 * T) * 1e6, "us_per_tick_total": t_all / (4 * T) * 1e6}
    ekf.close()
print(json.dumps(out))
