#!/usr/bin/env python
"""Developer aid: one training step out of a `rocprofv3 --kernel-trace --output-format csv` trace -- every launch between two
`adam_step` launches (the last complete step of the run), with start, duration, gap to the previous launch and workgroup count, and
the sums per kernel.  Usage: python tools/step_timeline.py <..._kernel_trace.csv> [marker-kernel-substring]"""
import csv
import re
import sys
from collections import defaultdict

path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "adam_step"
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], grid // max(wg, 1)))
rows.sort()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
if len(marks) < 3:
    sys.exit("fewer than three '%s' launches in the trace" % marker)
a, b = marks[-2] + 1, marks[-1] + 1          # the last complete step, its own marker launch included


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"^void ", "", name)
    name = name.replace("sstem::", "")
    name = re.sub(r"at::native::|at::cuda::|c10::|std::", "", name)
    return name[:90]


step = rows[a:b]
t0 = step[0][0]
busy = sum(e - s for s, e, _, _ in step)
span = step[-1][1] - t0
print("one step: %d launches, span %.1f us, kernel time %.1f us, gaps %.1f us" % (len(step), span / 1e3, busy / 1e3, (span - busy) / 1e3))
print("columns: launch index, start (us), duration (us), gap before (us), workgroups, kernel")
prev_end = t0
per = defaultdict(lambda: [0, 0.0])
for i, (s, e, n, g) in enumerate(step):
    print("%4d %9.1f %8.1f %6.1f %7d  %s" % (i, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, g, short(n)))
    prev_end = max(prev_end, e)
    per[short(n)][0] += 1
    per[short(n)][1] += (e - s) / 1e3
print("\nper kernel (launches, total us, share of the span):")
for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print("%4d %9.1f %5.1f %%  %s" % (c, t, 100.0 * t * 1e3 / span, n))
