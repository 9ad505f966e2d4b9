#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats directory: top kernels by total time, with shortened names.
Usage: python tools/kernel_stats_top.py <rocprof_out_dir> [n]"""
import csv
import glob
import re
import sys

d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print("total kernel time %.2f ms over %d launches, %d distinct kernels" % (tot / 1e6, calls, len(rows)))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:n]:
    name = r["Name"]
    name = name.replace("(anonymous namespace)::", "")      # before cutting the argument list at the first "("
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    name = name.replace("at::native::", "")
    print("  %5.1f %%  %6s calls  avg %9.1f us  %s" % (100 * float(r["TotalDurationNs"]) / tot, r["Calls"],
                                                     float(r["AverageNs"]) / 1e3, name[:110]))
