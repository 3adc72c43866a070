"""Per-workgroup timeline of the workgroup-cooperative tick (kw_tick<double, step>, BASELINE cfg 2: 4 096 fp64 filters, a correction
on every tick) from s_memtime stamps in the diagnostic build (make -C quadrotor_landing_amd/csrc dbg; QLE_LIB points at it).
Thread 0 of every workgroup stamps; medians over the 64 workgroups of the last launch, in shader-clock cycles."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quadrotor_landing_amd as qla  # noqa: E402
from bench import CFG3  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dtype = sys.argv[2] if len(sys.argv) > 2 else "f64"
cfg = dict(CFG3, update_freq=100.0, measurement_freq=100.0, limit_measurement_freq=0)
ekf = qla.BatchedRelativePoseEKF(B, dtype, **cfg)
T = 20
seq = ekf.make_inputs(T, np.ones(T, np.uint8))
ekf.synth_generate(seq, seed=2)
ekf.run(seq, 0, 40); ekf.synchronize()
ekf.timer_begin(); ekf.run(seq, 0, 400); ms = ekf.timer_end()
print(f"B={B} {dtype}: {ms / 400 * 1e3:.2f} us per tick (HIP-event period, instrumented build), policy {ekf.policy()}")
ekf.run(seq, 0, 1); ekf.synchronize()
L = C.CDLL(os.environ["QLE_LIB"])
NS = 128
nw = min(4096, (B // 64) * (4 if B <= 8192 else 1))
buf = (C.c_ulonglong * (nw * NS))()
fn = getattr(L, "qle_debug_clocks_kw_" + ("double" if dtype == "f64" else "float"))
assert fn(buf, nw * NS) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(nw, NS).astype(np.int64)
names = ["entry", "x, u (, z) arrived in the scalar wave", "predict_scalar done", "barrier 1 passed, P loads arrived", "predict_P done",
         "barrier 2 passed", "factor S = L D L^T done", "barrier 3 passed", "update_P done (stores issued)", "barrier 4 passed",
         "injection done", "end"]
d = t[:, :12] - t[:, :1]
end = np.median(d[:, 11])
print(f"median entry->end {end:.0f} cycles")
prev = 0
for k in range(12):
    m = np.median(d[:, k])
    print(f"  {k:2d} {names[k]:42s} {m:8.0f}  (+{m - prev:6.0f})  {m / end * 100:5.1f} %")
    prev = m
ekf.close()
