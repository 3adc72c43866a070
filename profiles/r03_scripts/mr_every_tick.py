"""The multirate filter with a tag pose on EVERY tick (the shipped hardware configuration's cadence: measurement_freq = update_freq =
100 Hz, 150 ms camera latency = 15 ticks, relative_pose_EKF_hardware.yaml): every launch is k_step_mr, every chain starts from the anchor
one tick before the measurement's entry.  HIP-event period per tick.
    python profiles/r03_scripts/mr_every_tick.py [batch] [f32|f64] [step delay in ticks]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quadrotor_landing_amd as qla  # noqa: E402
from bench import CFG3  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
step = int(sys.argv[3]) if len(sys.argv) > 3 else 15
cfg = dict(CFG3, update_freq=100.0, measurement_freq=100.0, limit_measurement_freq=0, multirate_ekf=1, dynamic_meas_delay=1,
           measurement_delay=0.150, measurement_delay_max=0.350, dyn_measurement_delay_offset=0.085)
ekf = qla.BatchedRelativePoseEKF(B, dtype, **cfg)
T = 120
seq = ekf.make_inputs(T, np.ones(T, np.uint8))
ekf.set_uniform_measurement_age(step / cfg["update_freq"] - cfg["dyn_measurement_delay_offset"])
ekf.synth_generate(seq, seed=3, meas_delay_ticks=step)
ekf.run(seq, 0, T); ekf.synchronize()
ekf.timer_begin(); ekf.run(seq, 0, T); ms = ekf.timer_end()
print(f"B={B} {dtype}, tag pose on every tick, step delay {step}: {ms / T * 1e3:.2f} us per tick, {B * T / ms * 1e3:.3e} ticks/s, non-finite {ekf.count_nonfinite()}")
ekf.close()
