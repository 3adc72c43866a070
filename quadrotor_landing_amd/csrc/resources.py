#!/usr/bin/env python3
"""Per-kernel VGPR / AGPR / SGPR / scratch / LDS / occupancy table for the engine's kernels, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks (cross-compiles, no GPU).

    python resources.py [regex-filter]          # all translation units, both dtypes, the Makefile's flags
"""
import concurrent.futures
import os
import re
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
flt = sys.argv[1] if len(sys.argv) > 1 else ""
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=max-memory-clause",
         "-fno-hip-fp32-correctly-rounded-divide-sqrt", "-mllvm", "-amdgpu-remove-redundant-endcf=0", "-mllvm", "-amdgpu-kernarg-preload-count=14", "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null"]
jobs = [("ekf_capi.hip", None)] + [(f"{tu}.hip", t) for tu in ("tu_predict", "tu_step", "tu_quad", "tu_misc", "tu_compact") for t in ("float", "double")]


def run(job):
    src, t = job
    cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + ([f"-DQLE_TU_T={t}"] if t else []) + (["-mllvm", "-disable-vector-combine"] if src == "tu_misc.hip" else []) + \
        (["-mllvm", "-disable-machine-licm"] if (src, t) in (("tu_misc.hip", "double"), ("tu_compact.hip", "double")) else []) + [src]
    return subprocess.run(cmd, cwd=here, capture_output=True, text=True).stderr


rows, cur = [], {}
with concurrent.futures.ThreadPoolExecutor(8) as ex:
    for out in ex.map(run, jobs):
        for line in out.splitlines():
            m = re.search(r"remark:\s+.*?(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
            if not m:
                continue
            k, v = m.group(1).replace(" ", ""), m.group(2)
            if k == "FunctionName":
                cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
                rows.append(cur)
            else:
                cur[k] = v
seen = set()
for r in sorted(rows, key=lambda r: r["name"]):
    name = re.sub(r"\(.*", "", r["name"]).replace("void qle::", "")
    if name in seen or (flt and not re.search(flt, name)):
        continue
    seen.add(name)
    print(f"{name:64s} V={r.get('VGPRs')} A={r.get('AGPRs')} S={r.get('TotalSGPRs')} scratch={r.get('ScratchSize[bytes/lane]')} "
          f"lds={r.get('LDSSize[bytes/block]')} occ={r.get('Occupancy[waves/SIMD]')}")
