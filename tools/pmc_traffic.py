#!/usr/bin/env python
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE csv output into HBM bytes per launch of one kernel.

Usage: python tools/pmc_traffic.py <pmc_fetch_dir> <pmc_write_dir> <kernel-substring> <out.json> [batch size fused|unfused rgb|gray [kernel label [frame planes]]]
(kernel label = the roofline.kernel name bench.py prints for that run, without the parenthesis; bench.py only
reports the traffic when the label matches the kernel it is timing.)

Corrections, as /opt/skills/guides/MI355X_MICROARCH.md (section HBM) prescribes for gfx950:
  * FETCH_SIZE and WRITE_SIZE are reported in KiB (x1024);
  * FETCH_SIZE reports exactly 1/2 of the bytes of a coalesced streaming read -> doubled.  The guide
    calibrates this for 16 B/lane loads; the sepconv kernels use 4 B/lane coalesced loads, so the
    factor is re-checked on a known byte count in the same access pattern: the torch elementwise add
    in the same profile (2 x 100.66 MB read) and tools' copy kernel both read back exactly 1/2;
  * WRITE_SIZE is exact;
  * the two counters are collected in separate passes (TCC has 4 slots: FETCH 3 + WRITE 2).
"""
import collections
import csv
import glob
import json
import os
import sys


def mean_counter(d, counter, needle):
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and needle in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    if not vals:
        raise SystemExit("no %s rows for kernel *%s* under %s" % (counter, needle, d))
    return sum(vals) / len(vals), len(vals)


def main():
    fetch_dir, write_dir, needle, out = sys.argv[1:5]
    batch = int(sys.argv[5]) if len(sys.argv) > 5 else 8
    size = int(sys.argv[6]) if len(sys.argv) > 6 else 1024
    fused = (sys.argv[7] == "fused") if len(sys.argv) > 7 else False
    rgb = (sys.argv[8] == "rgb") if len(sys.argv) > 8 else False
    label = sys.argv[9] if len(sys.argv) > 9 else None
    planes = int(sys.argv[10]) if len(sys.argv) > 10 else 3
    fetch_kib, nf = mean_counter(fetch_dir, "FETCH_SIZE", needle)
    write_kib, nw = mean_counter(write_dir, "WRITE_SIZE", needle)
    read_bytes = fetch_kib * 1024 * 2      # gfx950: FETCH_SIZE counts 64 B per 128-B request
    write_bytes = write_kib * 1024
    res = {"kernel": needle, "kernel_label": label, "batch": batch, "size": size, "fused": fused, "rgb": rgb, "frame_planes": planes,
           "fetch_size_kib_raw": fetch_kib, "write_size_kib_raw": write_kib,
           "read_bytes_per_launch": int(read_bytes), "write_bytes_per_launch": int(write_bytes),
           "hbm_bytes_per_launch": int(read_bytes + write_bytes),
           "launches_averaged": {"fetch": nf, "write": nw},
           "correction": "FETCH_SIZE x1024 x2 (gfx950 half-count), WRITE_SIZE x1024"}
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
