#!/usr/bin/env python3
"""Condense rocprofv3 output directories (kernel-trace / --stats / --pmc CSVs) into the small summaries committed under profiles/.

    python profiles/summarize.py <rocprof dir> <out.md> [label]                      kernel stats / launch resources / PMC means
    python profiles/summarize.py --traffic <fetch dir> <write dir> <key> <kernel regex> <traffic.json>
                                                                                      HBM bytes per launch of one kernel from the separate
                                                                                      FETCH_SIZE and WRITE_SIZE passes, merged into traffic.json

Every file is stamped with the head the measurement was taken on (QLE_HEAD_SHA, exported by the profiling script: the GPU box has no .git).
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section), so HBM bytes = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

SHA = os.environ.get("QLE_HEAD_SHA", "unknown")


def counter_means(src):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            key = (r["Kernel_Name"], r["Counter_Name"])
            agg[key][0] += 1
            agg[key][1] += float(r["Counter_Value"])
    return agg


def traffic(fetch_dir, write_dir, key, kernel_re, out_json):
    rx = re.compile(kernel_re)
    got = {}
    for name, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
        for (k, c), (n, s) in counter_means(d).items():
            if c == name and rx.search(k):
                prev = got.get(name)
                if prev is None or n > prev[1]:
                    got[name] = (s / n, n, k)
    if len(got) != 2:
        sys.exit(f"traffic: counters for /{kernel_re}/ not found in {fetch_dir} / {write_dir}: {got}")
    fetch_kib, write_kib = got["FETCH_SIZE"][0], got["WRITE_SIZE"][0]
    tj = {"per_launch": {}}
    if os.path.exists(out_json):
        tj = json.load(open(out_json))
        if tj.get("sha") != SHA:      # measurements of another head do not mix
            tj = {"per_launch": {}}
    tj["sha"] = SHA
    tj["units"] = ("bytes per launch; fabric_bytes = 2 x FETCH_SIZE + WRITE_SIZE (rocprofv3 reports KiB; FETCH_SIZE doubled per the gfx950 note): what "
                   "the L2s exchange with the fabric, Infinity-Cache hits included -- HBM bytes only where the state cannot stay on die")
    tj["per_launch"][key] = {"kernel": got["FETCH_SIZE"][2][:100], "fetch_size_kib": fetch_kib, "write_size_kib": write_kib,
                             "dispatches": [got["FETCH_SIZE"][1], got["WRITE_SIZE"][1]],
                             "fabric_bytes": 2 * fetch_kib * 1024 + write_kib * 1024}
    json.dump(tj, open(out_json, "w"), indent=1)
    print("traffic", key, tj["per_launch"][key])


def main():
    if sys.argv[1] == "--traffic":
        return traffic(*sys.argv[2:7])
    src, dst = sys.argv[1], sys.argv[2]
    label = sys.argv[3] if len(sys.argv) > 3 else src
    lines = [f"# rocprofv3 summary: {label}", "", f"head: `{SHA}`", ""]
    for f in sorted(glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)):
        rows = list(csv.DictReader(open(f)))
        lines += ["## kernel stats (`--kernel-trace --stats`)", "", "| kernel | calls | avg ns | min ns | max ns | total ns | % |", "|---|---|---|---|---|---|---|"]
        setup = ("k_synth", "k_seed", "k_rmse", "k_count_nonfinite", "fillBuffer", "copyBuffer", "k_fill_i32")
        rows.sort(key=lambda r: any(t in r["Name"] for t in setup))   # the tick kernels first; set-up / read-out kernels (outside every timed region) after them
        for r in rows:
            tag = " (set-up / read-out, untimed)" if any(t in r["Name"] for t in setup) else ""
            lines.append(f"| `{r['Name'][:100]}`{tag} | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {r['TotalDurationNs']} | {r['Percentage']} |")
        lines.append("")
    for f in sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)):
        rows = list(csv.DictReader(open(f)))
        seen = collections.OrderedDict()
        for r in rows:
            k = r["Kernel_Name"][:100]
            if k not in seen:
                seen[k] = r
        lines += ["## per-kernel launch resources (first dispatch)", "", "| kernel | VGPR | AGPR | SGPR | scratch B | LDS B | workgroup | grid |", "|---|---|---|---|---|---|---|---|"]
        for k, r in seen.items():
            lines.append(f"| `{k}` | {r.get('VGPR_Count')} | {r.get('Accum_VGPR_Count')} | {r.get('SGPR_Count')} | {r.get('Scratch_Size')} | {r.get('LDS_Block_Size')} | {r.get('Workgroup_Size_X')} | {r.get('Grid_Size_X')} |")
        lines += ["", "(rocprofv3's `VGPR_Count` on gfx950 reads about HALF the registers the compiler allocates per lane -- e.g. 92 for the 179-register "
                  "`k_predict<float>` -- and `Accum_VGPR_Count` likewise; the allocation figures quoted in DESIGN.md come from "
                  "`quadrotor_landing_amd/csrc/resources.py`, i.e. hipcc's `-Rpass-analysis=kernel-resource-usage`.)", ""]
    agg = counter_means(src)
    if agg:
        lines += ["## PMC counters", "", "| kernel | counter | dispatches | mean value per dispatch |", "|---|---|---|---|"]
        for (k, c), (n, s) in sorted(agg.items()):
            lines.append(f"| `{k[:100]}` | {c} | {n} | {s / n:.1f} |")
        lines.append("")
    open(dst, "w").write("\n".join(lines) + "\n")
    print("wrote", dst)


if __name__ == "__main__":
    main()
