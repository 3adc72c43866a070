"""Per-wave timeline of the multirate correcting tick (k_step_mr<float>) from s_memtime stamps in the diagnostic build
(`make -C quadrotor_landing_amd/csrc dbg` -> quadrotor_landing_amd/libqle_dbg.so, loaded through QLE_LIB).  cfg3mr schedule:
65 536 fp32 filters, tag poses every 14th tick arriving 12 ticks late.  Prints the median over the waves of the last correcting
launch: when each stamp was reached (s_memtime ticks = shader clock cycles) and the spacing of the loop iterations."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quadrotor_landing_amd as qla  # noqa: E402
from bench import CFG3  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
step = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dtype = sys.argv[3] if len(sys.argv) > 3 else "f32"
cfg = dict(CFG3, multirate_ekf=1, dynamic_meas_delay=1, measurement_delay=0.030, measurement_delay_max=0.200, dyn_measurement_delay_offset=0.005)
ekf = qla.BatchedRelativePoseEKF(B, dtype, **cfg)
T = 14 * 12
thm = np.zeros(T, np.uint8); thm[13::14] = 1
seq = ekf.make_inputs(T, thm)
ekf.set_uniform_measurement_age(step / cfg["update_freq"] - cfg["dyn_measurement_delay_offset"])
ekf.synth_generate(seq, seed=3, meas_delay_ticks=step)
ekf.run(seq, 0, T); ekf.synchronize()
ekf.timer_begin(); ekf.run(seq, 0, T); ms = ekf.timer_end()
print(f"B={B}, step delay {step}: {ms / T * 1e3:.2f} us per tick over the schedule (HIP-event period, instrumented build)")
ekf.run(seq, 0, 14); ekf.synchronize()          # ends on a correcting tick: its stamps are the ones left in the buffer
L = C.CDLL(os.environ["QLE_LIB"])
NS = 128
nw = min(B // 64, 4096)
buf = (C.c_ulonglong * (nw * NS))()
rc = (L.qle_debug_clocks_float if dtype == "f32" else L.qle_debug_clocks_double)(buf, nw * NS)
assert rc == 0, rc
t = np.frombuffer(buf, dtype=np.uint64).reshape(nw, NS).astype(np.int64)
t0 = t[:, 0]
rel = lambda k: t[:, k] - t0
end = np.median(rel(7))
print(f"median entry->end {end:.0f} s_memtime ticks (shader clock); spread of wave entry times {np.percentile(t0, 99) - np.percentile(t0, 1):.0f} ticks (p1..p99)")
names = {1: "inputs + x arrived", 2: "chain start decided", 3: "chain state (checkpoint / anchor) arrived", 5: "correction begins", 6: "correction done, anchor stores issued", 7: "end"}
for k in (1, 2, 3, 5, 6, 7):
    d = rel(k)
    print(f"  {names[k]:46s} median {np.median(d):7.0f}  p10 {np.percentile(d, 10):7.0f}  p90 {np.percentile(d, 90):7.0f}  ({np.median(d) / end * 100:5.1f} %)")
# loop iterations: 8 + 2 j sample ready, 9 + 2 j predict done
nj = 0
while 9 + 2 * nj < NS and np.median(t[:, 9 + 2 * nj]) >= np.median(t0) and np.median(rel(9 + 2 * nj)) > 0 and np.median(rel(9 + 2 * nj)) <= end * 1.01:
    nj += 1
print(f"  loop iterations seen: {nj}")
prev = np.median(rel(3))
for j in range(nj):
    a, b = np.median(rel(8 + 2 * j)), np.median(rel(9 + 2 * j))
    print(f"    iter {j:2d}: IMU sample in registers at {a:7.0f} (+{a - prev:6.0f} after the previous predict), predict done (x and P) at {b:7.0f} (predict {b - a:6.0f})")
    prev = b
if dtype == "f64":   # phases of the last split predict (ekf_split.hpp)
    sb = (C.c_ulonglong * (nw * 8))()
    assert L.qle_debug_split_clocks_double(sb, nw * 8) == 0
    st = np.frombuffer(sb, dtype=np.uint64).reshape(nw, 8).astype(np.int64)
    d = np.diff(st[:, :4], axis=1)
    print("  last split predict of a wave: row r %.0f, row v %.0f, rows th/ab/wb + noise %.0f cycles (medians)" % tuple(np.median(d, axis=0)))
ekf.close()
