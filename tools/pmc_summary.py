#!/usr/bin/env python3
"""Summarises one rocprofv3 output directory into small csv files worth committing under profiles/.

    python tools/pmc_summary.py <rocprofv3 -d directory> <out prefix>

* `*counter_collection.csv` -> `<prefix>.pmc.csv`: per kernel and counter, the number of DISPATCHES (rows of one dispatch --
  one per XCD / SE instance of the counter -- are summed first), the sum and the mean per dispatch;
* `*kernel_trace.csv` -> `<prefix>.durations.csv`: per kernel, launches and total / mean / min / max duration in microseconds
  (with --pmc the kernels run serialised and slower: quote durations from a --stats pass, not from a counter pass).
Kernel names lose `void`, the anonymous namespace and their argument list (template arguments stay)."""
import collections
import csv
import glob
import os
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:80]


def main():
    d, prefix = sys.argv[1], sys.argv[2]
    per = collections.defaultdict(lambda: collections.defaultdict(float))  # (kernel, counter) -> dispatch -> value
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            per[(short(r.get("Kernel_Name", "?")), r.get("Counter_Name", "?"))][r.get("Dispatch_Id", "0")] += float(r.get("Counter_Value", 0) or 0)
    if per:
        with open(prefix + ".pmc.csv", "w") as o:
            o.write("kernel,counter,dispatches,sum,mean_per_dispatch\n")
            for (k, c), disp in sorted(per.items()):
                s = sum(disp.values())
                o.write('"%s",%s,%d,%.9g,%.9g\n' % (k, c, len(disp), s, s / len(disp)))
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[short(r.get("Kernel_Name", "?"))].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    if dur:
        with open(prefix + ".durations.csv", "w") as o:
            o.write("kernel,launches,total_us,mean_us,min_us,max_us\n")
            for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
                o.write('"%s",%d,%.3f,%.3f,%.3f,%.3f\n' % (k, len(v), sum(v), sum(v) / len(v), min(v), max(v)))


if __name__ == "__main__":
    main()
