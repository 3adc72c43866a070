#!/usr/bin/env python3
"""Condense a rocprofv3 output directory (kernel-trace / --stats / --pmc CSVs) into the small
summaries committed under profiles/.

    python profiles/summarize.py gpurun_out/prof1 profiles/r01_kernel_stats.md [label]
"""
import collections
import csv
import glob
import os
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    label = sys.argv[3] if len(sys.argv) > 3 else src
    lines = [f"# rocprofv3 summary: {label}", ""]
    for f in sorted(glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)):
        rows = list(csv.DictReader(open(f)))
        lines += ["## kernel stats (`--kernel-trace --stats`)", "", "| kernel | calls | avg ns | min ns | max ns | total ns | % |", "|---|---|---|---|---|---|---|"]
        for r in rows:
            lines.append(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {r['TotalDurationNs']} | {r['Percentage']} |")
        lines.append("")
    for f in sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)):
        rows = list(csv.DictReader(open(f)))
        seen = collections.OrderedDict()
        for r in rows:
            k = r["Kernel_Name"][:90]
            if k not in seen:
                seen[k] = r
        lines += ["## per-kernel launch resources (first dispatch)", "", "| kernel | VGPR | AGPR | SGPR | scratch B | LDS B | workgroup | grid |", "|---|---|---|---|---|---|---|---|"]
        for k, r in seen.items():
            lines.append(f"| `{k}` | {r.get('VGPR_Count')} | {r.get('Accum_VGPR_Count')} | {r.get('SGPR_Count')} | {r.get('Scratch_Size')} | {r.get('LDS_Block_Size')} | {r.get('Workgroup_Size_X')} | {r.get('Grid_Size_X')} |")
        lines.append("")
    for f in sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in rows:
            key = (r["Kernel_Name"][:90], r["Counter_Name"])
            agg[key][0] += 1
            agg[key][1] += float(r["Counter_Value"])
        lines += [f"## PMC counters ({os.path.basename(f)})", "", "| kernel | counter | dispatches | mean value per dispatch |", "|---|---|---|---|"]
        for (k, c), (n, s) in sorted(agg.items()):
            lines.append(f"| `{k}` | {c} | {n} | {s / n:.1f} |")
        lines.append("")
    open(dst, "w").write("\n".join(lines) + "\n")
    print("wrote", dst)


if __name__ == "__main__":
    main()
