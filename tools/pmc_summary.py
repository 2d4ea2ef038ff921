#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one directory per pass) into one table per kernel.

    python tools/pmc_summary.py gpurun_out/pmc > profiles/<round>/pmc_summary.txt

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  On gfx950 FETCH_SIZE counts 64 B per 128-B request
for wide coalesced reads, so the read side is doubled before it is compared with a byte count
(/opt/skills/guides/MI355X_MICROARCH.md, section HBM); WRITE_SIZE is exact for 16-B-per-lane stores.
"""
import collections
import csv
import glob
import os
import sys


def main(root):
    agg = collections.defaultdict(list)
    dur = collections.defaultdict(list)
    for path in sorted(glob.glob(os.path.join(root, "**", "*_counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"]
            if "nbk::" not in k:
                continue
            short = k.split("(")[0].replace("void ", "")
            agg[(short, r["Counter_Name"])].append(float(r["Counter_Value"]))
            dur[short].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    kernels = sorted({k for k, _ in agg})
    for k in kernels:
        print(f"== {k}   (mean duration under counters {sum(dur[k]) / len(dur[k]):.3f} ms, {len(dur[k])} samples)")
        vals = {c: sum(v) / len(v) for (kk, c), v in agg.items() if kk == k}
        for c in sorted(vals):
            print(f"   {c:24s} {vals[c]:.6g}")
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
            rd, wr = vals["FETCH_SIZE"] * 1024 * 2, vals["WRITE_SIZE"] * 1024
            print(f"   -> HBM-side traffic per launch: read {rd / 1e6:.2f} MB (FETCH_SIZE x 2, gfx950 correction) + "
                  f"write {wr / 1e6:.2f} MB = {(rd + wr) / 1e6:.2f} MB")
        if "SQ_WAVE_CYCLES" in vals and "SQ_ACTIVE_INST_ANY" in vals:
            wc = vals["SQ_WAVE_CYCLES"]
            print(f"   -> wave time: issuing {vals['SQ_ACTIVE_INST_ANY'] / wc:.1%}, issue-stalled "
                  f"{vals.get('SQ_WAIT_INST_ANY', 0) / wc:.1%}, parked (s_waitcnt/barrier) {vals.get('SQ_WAIT_ANY', 0) / wc:.1%}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc")
