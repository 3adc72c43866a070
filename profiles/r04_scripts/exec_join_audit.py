#!/usr/bin/env python3
"""Audit of gfx950 assembly (hipcc -S / -save-temps) for vector instructions the backend placed at a control-flow join BEFORE the
`s_or_b64 exec, exec, sN` that re-enables the lanes of the region that ends there.

Background (profiles/r04_tuning.md section 1): the register allocator's live-range split copies (v_accvgpr_write / v_mov / spills) can
land at the top of the join block of an `if`, in front of the EXEC restore.  When the `if` body was skipped with `s_cbranch_execz`
EXEC is 0 there (or, if some lanes took the body, only those lanes are enabled), so the copy does not happen for the other lanes; a
later reload under the restored EXEC then hands them an undefined value.  With the value being the per-lane filter index this is a
wild address: the GPU memory access faults of round 3 (block-form k_step_mr<double>) and of the max-ilp experiment of round 4.

For every label that is the target of an `s_cbranch_execz` the scan walks from the label to the first instruction that writes EXEC;
if that instruction is `s_or_b64 exec, exec, sN` (lanes are re-enabled there) the VALU / VMEM / DS instructions in between ran without
those lanes.  Pure re-materialisations
(v_mov of a literal) are harmless; copies of live registers (v_accvgpr_write aN, vM / v_accvgpr_read / v_mov vN, vM / scratch_store)
are reported as HAZARD.

    exec_join_audit.py file.s [kernel-name-regex]      exit code 1 if any HAZARD is found
"""
import re
import sys

text = open(sys.argv[1]).read().splitlines()
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
kernels, cur = {}, None
for ln in text:
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        cur = []
        kernels[m.group(1)] = cur
    elif cur is not None:
        cur.append(ln)

copy_re = re.compile(r"^(v_accvgpr_write_b32 a\d+, v\d+|v_accvgpr_read_b32 v\d+, a\d+|v_mov_b(32|64)_e32 v(\[\d+:\d+\]|\d+), (v|a)(\[\d+:\d+\]|\d+)|scratch_store|scratch_load|v_accvgpr_mov)")
vec_re = re.compile(r"^(v_|ds_|global_|scratch_|buffer_|flat_)")
exec_w = re.compile(r"^s_\w+ exec\b|^s_\w+saveexec|^s_mov_b64 exec|^s_cbranch|^s_branch|^s_endpgm|^s_setpc")
total = 0
for name, body in kernels.items():
    if pat and not pat.search(name):
        continue
    targets = set()
    for ln in body:
        m = re.match(r"\s*s_cbranch_execz (\.LBB\d+_\d+)", ln)
        if m:
            targets.add(m.group(1))
    haz, benign = [], 0
    for n, ln in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if not m or m.group(1) not in targets:
            continue
        found, other, widened = [], 0, False
        for k in range(n + 1, len(body)):
            s = body[k].strip()
            if not s or s.startswith(";"):
                continue
            if re.match(r"^s_or_b64 exec, exec,", s):
                widened = True      # lanes are re-enabled HERE: everything collected above ran without them
                break
            if re.match(r"^\.LBB", s) or exec_w.match(s):
                break
            if vec_re.match(s):
                if copy_re.match(s):
                    found.append((m.group(1), k, s))
                else:
                    other += 1
        if widened:
            haz += found
            benign += other
    total += len(haz)
    if haz:
        short = name[:70]
        print(f"{short}: {len(haz)} register copies under a stale EXEC at {len(set(h[0] for h in haz))} join(s); {benign} other vector instructions there")
        for lab, k, s in haz[:8]:
            print(f"    {lab} +{k}: {s}")
sys.exit(1 if total else 0)
