#!/usr/bin/env python3
"""Per-kernel durations and the idle gaps between consecutive kernels of a rocprofv3 --kernel-trace csv (one stream's timeline)."""
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
rows.sort()
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
for i, (s, e, n) in enumerate(rows):
    dur[n].append((e - s) / 1e3)
    if i: gap[rows[i - 1][2][:28] + " -> " + n[:28]].append((s - rows[i - 1][1]) / 1e3)
only = sys.argv[2] if len(sys.argv) > 2 else ""   # optional: only kernels / gaps whose name contains this substring
dur = {k: v for k, v in dur.items() if only in k}
gap = {k: v for k, v in gap.items() if only in k}
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print("%-50s n %5d  mean %8.2f us" % (n[:50], len(v), sum(v) / len(v)))
for n, v in sorted(gap.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print("gap %-60s n %5d  mean %8.2f us" % (n, len(v), sum(v) / len(v)))
