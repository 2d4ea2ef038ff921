#!/usr/bin/env python3
"""FAST whole-set step, pairs form (NB_FAST_PAIRS=1; waves per workgroup 8, 4, 2, 1) against the ordered scalar-load fold
(NB_FAST_PAIRS=0), ms per step by size: where make_plan's bounds for the pairs form come from (profiles/r03/pairs_sizes.log).
Usage: pairs_sizes.py [N ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nenbody_amd as nb  # noqa: E402
nb._lib.use_library(nb._lib.LEGACY_LIB_PATH)   # this script names launch shapes only the legacy build holds (make -C nenbody_amd/csrc legacy)
nb.reload_env()  # tools/ read the NB_* kernel-form knobs; a host that merely loads the library does not (nb_diag_enable_env)

sizes = [int(x) for x in sys.argv[1:]] or [8192, 16384, 24576, 32768, 49152, 65536, 81920, 98304, 114688, 131072, 196608, 262144]
for n in sizes:
    pos, vel = nb.init_state(n, 1234)
    row = []
    for pairs, w in (("0", "0"), ("1", "8"), ("1", "4"), ("1", "2"), ("1", "1")):
        os.environ["NB_FAST_PAIRS"] = pairs
        os.environ["NB_FAST_PAIRS_W"] = w
        if w != "0" and n <= 262144 and n / (256 * int(w)) * n * 12 > 1.2e9:   # one tile's rows beyond 1.2 GB: skip (larger sets go in chunks)
            row.append(float("nan"))
            continue
        nb.reload_env()
        steps = max(10, min(400, int(3e11 / (float(n) * n))))
        with nb.Scene(pos, vel, nb.default_params(mode=nb.NB_MODE_FAST)) as sc:
            sc.step_n(max(steps, 20))
            sc.sync()
            best = 1e9
            for _ in range(4):
                t0 = time.perf_counter()
                sc.step_n(steps)
                sc.sync()
                best = min(best, (time.perf_counter() - t0) / steps * 1e3)
        row.append(best)
    print(f"N={n:7d}: ordered {row[0]:8.4f} ms   pairs W=8 {row[1]:8.4f}  W=4 {row[2]:8.4f}  W=2 {row[3]:8.4f}  W=1 {row[4]:8.4f}   best/ordered "
          f"{min(x for x in row[1:] if x == x) / row[0]:.3f}", flush=True)
os.environ.pop("NB_FAST_PAIRS", None)
os.environ.pop("NB_FAST_PAIRS_W", None)
nb.reload_env()
