#!/usr/bin/env python
"""MFMA-pipe utilisation of one kernel from two rocprofv3 --pmc passes (SQ counters, GRBM_GUI_ACTIVE).

Usage: python tools/mfma_util.py <kernel-substring> <sq_pass_dir> <grbm_pass_dir> [label]

  busy        = SQ_VALU_MFMA_BUSY_CYCLES               (cycles, summed over all SIMDs; = issue cycles x N_mfma)
  available   = GRBM_GUI_ACTIVE / 8 x 1024             (rocprofv3 reports the sum over the 8 XCDs; 256 CUs x 4 SIMDs)
  utilisation = busy / available
  clock       = GRBM_GUI_ACTIVE / 8 / dispatch time    (MI355X_MICROARCH.md, "DVFS give-back")
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (same guide): reported as shares of SQ_WAVE_CYCLES.
"""
import collections
import csv
import glob
import os
import sys


def collect(d, needle):
    acc, dur, seen = collections.defaultdict(list), [], set()
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if needle not in r["Kernel_Name"]:
                continue
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    if not dur:
        raise SystemExit("no dispatches of *%s* under %s" % (needle, d))
    return {k: sum(v) / len(v) for k, v in acc.items()}, sum(dur) / len(dur), len(dur)


def main():
    needle, sq_dir, grbm_dir = sys.argv[1:4]
    label = sys.argv[4] if len(sys.argv) > 4 else needle
    sq, dur_sq, n_sq = collect(sq_dir, needle)
    gr, dur_gr, n_gr = collect(grbm_dir, needle)
    gui = gr["GRBM_GUI_ACTIVE"]
    avail = gui / 8.0 * 1024.0
    busy = sq["SQ_VALU_MFMA_BUSY_CYCLES"]
    wc = sq.get("SQ_WAVE_CYCLES", 0.0)
    print("%s" % label)
    print("   dispatches %d / %d, mean duration %.1f us (SQ pass) / %.1f us (GRBM pass), clock %.2f GHz" %
          (n_sq, n_gr, dur_sq / 1e3, dur_gr / 1e3, gui / 8.0 / dur_gr))
    print("   MFMA instructions %.4g, MFMA busy cycles %.4g of %.4g available SIMD-cycles -> MFMA pipe %.1f %% busy" %
          (sq.get("SQ_INSTS_MFMA", float("nan")), busy, avail, 100.0 * busy / avail))
    if wc:
        print("   of the wave-cycles: issuing %.1f %%, waiting on an instruction (pipe / dependency) %.1f %%, parked "
              "(s_waitcnt / barrier) %.1f %%" % (100 * sq.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * sq.get("SQ_WAIT_INST_ANY", 0) / wc,
                                                 100 * sq.get("SQ_WAIT_ANY", 0) / wc))


if __name__ == "__main__":
    main()
