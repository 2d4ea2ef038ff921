#!/usr/bin/env python3
"""ms per FAST step at N (default 131072) in the pairs form (NB_FAST_PAIRS=1), best of 5 x STEPS steps.  Usage: pairs_time.py [N [STEPS]]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("NB_FAST_PAIRS", "1")
import nenbody_amd as nb  # noqa: E402
nb.reload_env()  # tools/ read the NB_* kernel-form knobs; a host that merely loads the library does not (nb_diag_enable_env)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
pos, vel = nb.init_state(n, 1234)
with nb.Scene(pos, vel, nb.default_params(mode=nb.NB_MODE_FAST)) as sc:
    sc.step_n(60)
    sc.sync()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        sc.step_n(steps)
        sc.sync()
        best = min(best, (time.perf_counter() - t0) / steps * 1e3)
print(f"N={n} {nb._lib.planned_kernels(nb.default_params(mode=nb.NB_MODE_FAST), n, n)[0]}: {best:.4f} ms/step", flush=True)
