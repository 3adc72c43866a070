#!/usr/bin/env python3
"""Per-kernel VGPR / AGPR / SGPR / scratch / occupancy table for the engine's kernels,
from hipcc's -Rpass-analysis=kernel-resource-usage remarks (cross-compiles, no GPU)."""
import os
import re
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
src = sys.argv[1] if len(sys.argv) > 1 else "ekf_capi.hip"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-slp-vectorize",
       "-mllvm", "-amdgpu-sched-strategy=max-memory-clause",
       "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", src]
out = subprocess.run(cmd, cwd=here, capture_output=True, text=True).stderr
rows, cur = [], {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1).replace(" ", ""), m.group(2)
    if k == "FunctionName":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    else:
        cur[k] = v
for r in rows:
    name = re.sub(r"\(.*", "", r["name"]).replace("void qle::", "")
    if flt and not re.search(flt, name):
        continue
    print(f"{name:58s} V={r.get('VGPRs')} A={r.get('AGPRs')} S={r.get('TotalSGPRs')} scratch={r.get('ScratchSize[bytes/lane]')} spill={r.get('VGPRsSpill')} occ={r.get('Occupancy[waves/SIMD]')}")
