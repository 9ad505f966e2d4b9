#!/usr/bin/env python
"""LDS bank-conflict share per kernel from a `rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv` directory:
kernels by LDS-active cycles, with the share of them that were conflicts.  Usage: python tools/pmc_lds_by_kernel.py <dir>"""
import collections, csv, glob, os, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(os.path.join(sys.argv[1], "**", "*_counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("(anonymous namespace)::", "")[:110]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); cnt[k] += 1
rows = sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0.0))
print("%9s %14s %14s %7s  kernel" % ("launches", "LDS active", "conflicts", "share"))
for k, v in rows[:25]:
    a, c = v.get("SQ_LDS_IDX_ACTIVE", 0.0), v.get("SQ_LDS_BANK_CONFLICT", 0.0)
    if a <= 0: continue
    print("%9d %14.0f %14.0f %6.1f%%  %s" % (cnt[k], a, c, 100.0 * c / a, k))
