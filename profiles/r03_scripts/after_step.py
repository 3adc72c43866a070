"""Kernel durations by distance from a correcting tick, from a rocprofv3 --kernel-trace CSV:
   python3 profiles/r03_scripts/after_step.py <dir with *_kernel_trace.csv> <kernel substring, e.g. k_step_mr>
The tick kernels of one stream run back to back; a predict tick that follows a correcting tick meets the caches as that tick left them."""
import csv, glob, statistics as st, sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if "qle::k" in r["Kernel_Name"] and "synth" not in r["Kernel_Name"] and "seed" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
print("launches of %s: %d" % (sys.argv[2], len(idx)))
print("| distance | median ns | mean ns | kernel |\n|---|---|---|---|")
for d in range(-2, 9):
    v = [int(rows[i + d]["End_Timestamp"]) - int(rows[i + d]["Start_Timestamp"]) for i in idx if 0 <= i + d < len(rows)]
    name = rows[idx[0] + d]["Kernel_Name"][10:48] if 0 <= idx[0] + d < len(rows) else ""
    print("| %d | %d | %d | `%s` |" % (d, st.median(v), st.mean(v), name))
