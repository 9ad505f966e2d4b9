#!/usr/bin/env python
"""Print mean counter values (and dispatch duration) per kernel from rocprofv3 --pmc csv output dirs."""
import collections
import csv
import glob
import os
import sys


def main():
    needle = sys.argv[1]
    for d in sys.argv[2:]:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            acc = collections.defaultdict(list)
            dur = []
            seen = set()
            for r in csv.DictReader(open(f)):
                if needle not in r["Kernel_Name"]:
                    continue
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                if r["Dispatch_Id"] not in seen:
                    seen.add(r["Dispatch_Id"])
                    dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            if dur:
                print("%s: %d dispatches, mean duration %.1f us" % (d, len(dur), sum(dur) / len(dur) / 1e3))
            for k, v in sorted(acc.items()):
                print("   %-32s %16.1f" % (k, sum(v) / len(v)))


if __name__ == "__main__":
    main()
