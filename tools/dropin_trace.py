#!/usr/bin/env python3
"""A few hundred drop-in calls at N (default 100) for a rocprofv3 --kernel-trace run: where a call's microseconds go (kernel
durations and the gaps between the three kernels of a call).  dropin_trace.py [N [boids]]; with --analyse DIR it reads the
trace back and prints the per-call breakdown."""
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == "--analyse":
    import csv

    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[len(rows) // 2:]   # the second half: warm
    names = [r["Kernel_Name"].split("(")[0].split("::")[-1][:28] for r in rows]
    per = {}
    for k, r in enumerate(rows):
        per.setdefault(names[k], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for nme, d in per.items():
        print(f"  {nme:30s} {len(d):5d} launches, mean {sum(d) / len(d):6.2f} us")
    gaps = {}
    for k in range(1, len(rows)):
        g = (int(rows[k]["Start_Timestamp"]) - int(rows[k - 1]["End_Timestamp"])) / 1e3
        gaps.setdefault(f"{names[k - 1]} -> {names[k]}", []).append(g)
    for nme, d in gaps.items():
        d.sort()
        print(f"  gap {nme:50s} median {d[len(d) // 2]:6.2f} us")
    sys.exit(0)

import numpy as np  # noqa: E402

import nenbody_amd as nb  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
boids = len(sys.argv) > 2
pos, vel = nb.init_state(n, 1234)
inst = np.zeros((n, 4, 4), np.float32)
p, v = pos.copy(), vel.copy()
op, ov = np.zeros_like(p), np.zeros_like(v)
for _ in range(400):
    (nb.update_instance_boids if boids else nb.update_instance_nbody)(inst, p, op, v, ov)
print("done")
