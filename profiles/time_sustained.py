#!/usr/bin/env python3
"""SUSTAINED tick time of the streamed per-tick path: 0.6 s of back-to-back ticks over a long generated input
sequence (about 6 GB resident, like bench.py), reported chunk by chunk so transients show.

    [QLE_NT=0|1|2] [QLE_REFRESH=R] python profiles/time_sustained.py <batch> predict|mixed

`mixed` = the cfg3 schedule (every 14th tick fused).  Found with this script (profiles/r01_tuning.md section 5): with
non-temporal loads AND stores on every tick a 36 MiB state starts at 9.1 us per predict and decays to 10.9 us within
~3 000 ticks, so numbers taken in the first 40 ms of a process were optimistic."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadrotor_landing_amd as qla  # noqa: E402
from bench import CFG3  # noqa: E402

B = int(sys.argv[1])
mixed = sys.argv[2] == "mixed"
ekf = qla.BatchedRelativePoseEKF(B, "f32", **CFG3)
T = max(140, int(4200 * 65536 / B) // 14 * 14)         # about 6 GB of inputs, whole measurement periods
if os.environ.get("QLE_SEQ_TICKS"):
    T = int(os.environ["QLE_SEQ_TICKS"])               # e.g. 14: inputs small enough to stay cached
thm = np.zeros(T, np.uint8)
if mixed:
    thm[13::14] = 1
seq = ekf.make_inputs(T, thm if mixed else None)
ekf.synth_generate(seq, seed=3)
ekf.synchronize()
n = max(140, int(1400 * 65536 / B) // 14 * 14)         # ticks per reported chunk
pos = 0
series = []
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.6:
    ekf.timer_begin(); ekf.run(seq, pos, n); ms = ekf.timer_end(); pos = (pos + n) % T
    series.append(ms / n * 1e3)
alg = ekf.algorithmic_bytes(0) if not mixed else (13 * ekf.algorithmic_bytes(0) + ekf.algorithmic_bytes(1)) / 14
first, last = series[0], float(np.mean(series[-5:]))
print(os.path.basename(os.environ.get("QLE_LIB", "main")), end=" ")
print("B=%d %s NT=%s chunk=%d ticks: first %.2f us (%.0f GB/s)  sustained %.2f us (%.0f GB/s)  series: %s" % (
    B, sys.argv[2], os.environ.get("QLE_NT", "auto"), n, first, alg / first / 1e3, last, alg / last / 1e3,
    " ".join("%.1f" % s for s in series[:: max(1, len(series) // 12)])), flush=True)
