#!/usr/bin/env python3
"""Audit of compiler-inserted scratch spills against the EXEC mask they run under (gfx950 assembly from hipcc -save-temps).

A VGPR spill store executed while EXEC is partial saves only the active lanes; a reload of the same slot executed under a WIDER
EXEC hands the other lanes whatever the slot held before.  That is harmless for a value only the active lanes use, and a wild
address when the value is a per-lane offset that a later wave-wide load adds to a buffer base.

The scan is linear over one kernel's text: s_and_saveexec / s_or_saveexec open a divergent region, `s_or_b64 exec, exec, sN`
closes it, so every instruction gets the stack of open regions.  A slot is flagged when some reload runs outside a region that
every store of that slot is inside of (the store cannot have covered the reload's lanes).

    spill_exec_audit.py file.s [kernel-name-regex]
"""
import re
import sys

text = open(sys.argv[1]).read().splitlines()
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
kernels, cur, name = {}, None, None
for ln in text:
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        name, cur = m.group(1), []
        kernels[name] = cur
    elif cur is not None:
        cur.append(ln)
        if "s_endpgm" in ln and ".end" in ln:
            cur = None

for name, body in kernels.items():
    if pat and not pat.search(name):
        continue
    stack, rid, slots = [], 0, {}
    saved = {}
    for n, ln in enumerate(body):
        s = ln.strip()
        m = re.match(r"s_(and|or|xor|andn2)_saveexec_b64 (s\[\d+:\d+\]|vcc)", s)
        if m:
            rid += 1
            stack.append((rid, m.group(2)))
            continue
        m = re.match(r"s_or_b64 exec, exec, (s\[\d+:\d+\]|vcc)", s)
        if m and stack:
            # closes the innermost region that saved into this register pair (the structurizer nests them properly)
            for k in range(len(stack) - 1, -1, -1):
                if stack[k][1] == m.group(1):
                    del stack[k:]
                    break
            else:
                stack.pop()   # the saved mask was moved to another register pair: the innermost region ends
            continue
        m = re.match(r"scratch_(store|load)_dword(x\d)? .*offset:(\d+)", s)
        if m and ("Spill" in s or "Reload" in s):
            kind, off = m.group(1), int(m.group(3))
            slots.setdefault(off, []).append((kind, n, tuple(r for r, _ in stack)))
    bad = []
    for off, acc in sorted(slots.items()):
        stores = [a for a in acc if a[0] == "store"]
        loads = [a for a in acc if a[0] == "load"]
        if not stores:
            continue
        common = set(stores[0][2])
        for st in stores[1:]:
            common &= set(st[2])
        for ld in loads:
            missing = common - set(ld[2]) - {1}   # region 1 is the wave-uniform "whole wave beyond the batch" exit
            if missing:
                bad.append((off, ld[1], sorted(missing), [st[1] for st in stores]))
                break
    print(f"{name[:60]}: {len(slots)} spill slots, {len(bad)} reloaded under a wider EXEC than any store")
    for off, ldn, miss, stn in bad[:12]:
        print(f"   slot {off}: reload at +{ldn} outside divergent region(s) {miss}; stores at {stn}")
