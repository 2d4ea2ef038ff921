#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one directory per pass) into one table per kernel.

    python tools/pmc_summary.py gpurun_out/pmc > profiles/<round>/pmc_summary.txt
    python tools/pmc_summary.py gpurun_out/pmc --json profiles/hbm_traffic.json --n 131072 --count 131072 --source profiles/<round>/

--json writes the HBM bytes per launch of every kernel (no hand transcription), stamped with the sha of the kernel
sources they were measured on; bench.py reports them as roofline.traffic and refuses a file with another stamp.

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  On gfx950 FETCH_SIZE counts 64 B per 128-B request
for wide coalesced reads, so the read side is doubled before it is compared with a byte count
(/opt/skills/guides/MI355X_MICROARCH.md, section HBM); WRITE_SIZE is exact for 16-B-per-lane stores.
"""
import argparse
import collections
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(root, json_path=None, n=None, count=None, source=None, shard=False):
    agg = collections.defaultdict(list)
    dur = collections.defaultdict(list)
    for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"]
            if "nbk::" not in k:
                continue
            short = k.split("(")[0].replace("void ", "")
            agg[(short, r["Counter_Name"])].append(float(r["Counter_Value"]))
            dur[short].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    kernels = sorted({k for k, _ in agg})
    traffic = {}
    for k in kernels:
        print(f"== {k}   (mean duration under counters {sum(dur[k]) / len(dur[k]):.3f} ms, {len(dur[k])} samples)")
        vals = {c: sum(v) / len(v) for (kk, c), v in agg.items() if kk == k}
        for c in sorted(vals):
            print(f"   {c:24s} {vals[c]:.6g}")
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
            rd, wr = vals["FETCH_SIZE"] * 1024 * 2, vals["WRITE_SIZE"] * 1024
            print(f"   -> HBM-side traffic per launch: read {rd / 1e6:.2f} MB (FETCH_SIZE x 2, gfx950 correction) + "
                  f"write {wr / 1e6:.2f} MB = {(rd + wr) / 1e6:.2f} MB")
            base = re.sub(r"<.*$", "", k.replace("nbk::", ""))
            # several instantiations of one kernel template in a run: keep the one with the most samples (the bench's own shape)
            if base not in traffic or traffic[base]["samples"] < len(dur[k]):
                traffic[base] = {"bytes_per_launch": int(rd) + int(wr), "read": int(rd), "write": int(wr), "samples": len(dur[k]),
                                 "kernel": k, "mean_ms_under_counters": sum(dur[k]) / len(dur[k])}
        if "SQ_WAVE_CYCLES" in vals and "SQ_ACTIVE_INST_ANY" in vals:
            wc = vals["SQ_WAVE_CYCLES"]
            print(f"   -> wave time: issuing {vals['SQ_ACTIVE_INST_ANY'] / wc:.1%}, issue-stalled "
                  f"{vals.get('SQ_WAIT_INST_ANY', 0) / wc:.1%}, parked (s_waitcnt/barrier) {vals.get('SQ_WAIT_ANY', 0) / wc:.1%}")
    if json_path and shard:
        # one rank's share of a multi-GPU job (count of n bodies): added to the existing file under "shards", same stamp required
        from nenbody_amd._lib import kernel_code_sha

        out = json.load(open(json_path))
        if out.get("code_sha") != kernel_code_sha():
            raise SystemExit(f"{json_path} is stamped for other device code: write the one-GPU shape first")
        out.setdefault("shards", {})[str(count)] = {"n": n, "count": count, "source": source or root, "kernels": traffic}
        json.dump(out, open(json_path, "w"), indent=1)
        print(f"added shard {count} to {json_path}: {sorted(traffic)}", file=sys.stderr)
    elif json_path:
        from nenbody_amd._lib import kernel_code_sha, kernel_source_sha

        out = {"_comment": "HBM-side bytes per launch from rocprofv3 --pmc passes (FETCH_SIZE x 1024 x 2 [gfx950 correction for wide "
                           "coalesced reads] + WRITE_SIZE x 1024). Written by tools/pmc_summary.py --json; bench.py reports the sum "
                           "over a step's kernels as roofline.traffic when code_sha (the device code of the benchmarked kernels in the "
                           "built library, nenbody_amd/_lib.py:kernel_code_sha) and the shape match; src_sha is informational.",
               "n": n, "count": count, "code_sha": kernel_code_sha(), "src_sha": kernel_source_sha(), "source": source or root, "kernels": traffic}
        json.dump(out, open(json_path, "w"), indent=1)
        print(f"wrote {json_path}: {sorted(traffic)}", file=sys.stderr)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("root", nargs="?", default="gpurun_out/pmc")
    ap.add_argument("--json")
    ap.add_argument("--n", type=int, default=131072)
    ap.add_argument("--count", type=int, default=131072)
    ap.add_argument("--source")
    ap.add_argument("--shard", action="store_true", help="add the kernels as the shape of one rank's share (--count of --n bodies) to an existing --json file")
    a = ap.parse_args()
    main(a.root, a.json, a.n, a.count, a.source, a.shard)
