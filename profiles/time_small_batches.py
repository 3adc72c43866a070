#!/usr/bin/env python3
"""Rows-across-lanes kernel vs one-lane-per-filter kernels at small batch sizes: us per tick (HIP events),
predict-only ticks and ticks where every filter corrects."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadrotor_landing_amd as qla
from bench import CFG3
N = 400
rows = []
for dtype in ("f32", "f64"):
    for B in (64, 1024, 4096, 8192, 16384, 32768):
        r = {"dtype": dtype, "batch": B}
        for fam, env in (("rows", str(1 << 40)), ("lanes", "0")):
            os.environ["QLE_ROWS_MAX"] = env
            for name, every in (("predict", 0), ("step", 1)):
                ekf = qla.BatchedRelativePoseEKF(B, dtype, **CFG3)
                thm = np.ones(14, np.uint8) * every
                seq = ekf.make_inputs(14, thm)
                ekf.synth_generate(seq, seed=3)
                ekf.run(seq, 0, 28); ekf.synchronize()
                ekf.timer_begin(); ekf.run(seq, 0, N); ms = ekf.timer_end()
                r[f"{fam}_{name}_us"] = round(ms / N * 1e3, 2)
                ekf.close()
        rows.append(r)
        print(json.dumps(r), flush=True)
