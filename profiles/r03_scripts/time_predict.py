import json, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quadrotor_landing_amd as qla
from bench import CFG3
B = int(sys.argv[1]); N = 300
ekf = qla.BatchedRelativePoseEKF(B, "f32", **CFG3)
seq = ekf.make_inputs(14, None); ekf.synth_generate(seq, seed=3)
ekf.run(seq, 0, 200); ekf.synchronize()
best = 1e9
for r in range(3):
    ekf.timer_begin(); ekf.run(seq, 0, N); ms = ekf.timer_end(); best = min(best, ms / N * 1e3)
pol = ekf.policy()
print(json.dumps(dict(B=B, env={k: os.environ.get(k) for k in ("QLE_NT", "QLE_REFRESH", "QLE_BLOCK", "QLE_SPLIT")}, us=round(best, 2), tbs=round(B * 1112 / best / 1e6, 2), policy=pol["state_policy"], block=pol["block"])))
