#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer boundary (never the bench `value`).

The reference hands inputs over by value (Eigen vectors); the C-ABI's qle_step(u, z, mask) takes
caller-owned host buffers, copies them to the device (pageable H2D + a pack kernel) and launches the
tick.  This measures that path end to end for cfg 3's batch, next to the HBM-resident path."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadrotor_landing_amd as qla  # noqa: E402
from bench import CFG3  # noqa: E402

B, T = 65536, 140
thm = np.zeros(T, np.uint8); thm[13::14] = 1
ekf = qla.BatchedRelativePoseEKF(B, "f32", **CFG3)
seq = ekf.make_inputs(T, thm)
ekf.synth_generate(seq, seed=0xE4F00003)
U = []; Z = []
for t in range(T):
    u, z, m = seq.download_tick(t)
    U.append(u); Z.append(z if thm[t] else None)
for t in range(14):
    ekf.step(U[t], Z[t])
ekf.synchronize()
t0 = time.perf_counter()
for t in range(T):
    ekf.step(U[t], Z[t])
ekf.synchronize()
dt_host = time.perf_counter() - t0
ekf.run(seq, 0, 14); ekf.synchronize()
t0 = time.perf_counter()
ekf.run(seq, 0, T)
ekf.synchronize()
dt_res = time.perf_counter() - t0
print(json.dumps({"batch": B, "ticks": T, "host_buffer_ticks_per_s": B * T / dt_host, "host_buffer_ms_per_tick": dt_host / T * 1e3,
                  "resident_ticks_per_s": B * T / dt_res, "resident_ms_per_tick": dt_res / T * 1e3,
                  "host_bytes_per_tick": B * 6 * 8 + B * 7 * 8 / 14,
                  "note": "host path = pageable hipMemcpyAsync of [B][6] fp64 (+[B][7] every 14th tick) + pack kernel + tick kernel, synchronised per chunk"}))
