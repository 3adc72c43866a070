"""Per-wave timeline of the fused tick (k_step<float>) from s_memtime stamps in an instrumented debug build (not in the tree:
the stamps sit in ekf_step_fused between its phases; QLE_LIB points at that build).  Prints, over all waves of the last launch, the
median time from kernel entry to each stamp."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quadrotor_landing_amd as qla  # noqa: E402
from bench import CFG3  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N = 200
ekf = qla.BatchedRelativePoseEKF(B, "f32", **CFG3)
seq = ekf.make_inputs(14, np.ones(14, np.uint8))
ekf.synth_generate(seq, seed=3)
ekf.run(seq, 0, 28); ekf.synchronize()
ekf.timer_begin(); ekf.run(seq, 0, N); ms = ekf.timer_end()
print(f"B={B}: {ms / N * 1e3:.2f} us per tick (HIP-event period, instrumented build)")
L = C.CDLL(os.environ["QLE_LIB"])
nw = min(B // 64, 4096)
buf = (C.c_ulonglong * (nw * 16))()
rc = L.qle_debug_clocks(buf, nw * 16)
assert rc == 0, rc
t = np.frombuffer(buf, dtype=np.uint64).reshape(nw, 16).astype(np.int64)
names = ["entry", "x arrived", "nominal done", "innovation+jacobians+noise done", "rows r,v arrived", "level 3 done", "last load arrived",
         "level 1 done", "factor done", "levels 2,0 done", "first P store issued", "half of the P stores issued", "last P store issued",
         "injection done, x stored", "end"]
d = t[:, :15] - t[:, :1]
span = np.median(d[:, 14])
print(f"median entry->end: {span:.0f} ticks of s_memtime; spread of wave entry times over the launch: {np.ptp(t[:, 0])} ticks")
for k in range(15):
    print(f"  {k:2d} {names[k]:34s} median {np.median(d[:, k]):8.0f}  p10 {np.percentile(d[:, k], 10):8.0f}  p90 {np.percentile(d[:, k], 90):8.0f}   ({np.median(d[:, k]) / span * 100:5.1f} % of the wave's life)")
