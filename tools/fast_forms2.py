#!/usr/bin/env python3
"""FAST whole-set launch forms at N bodies (default 131072), ms per step: the scalar-load kernel (default) against the LDS wave
form (NB_FAST_SL=0) and the pairs form (NB_FAST_PAIRS=1: every unordered pair once), planar data and NB_FORCE_3D=1; max |dv| of each
against the first after one step.  Usage: fast_forms2.py [N [STEPS]]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nenbody_amd as nb  # noqa: E402
nb.reload_env()  # tools/ read the NB_* kernel-form knobs; a host that merely loads the library does not (nb_diag_enable_env)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
pos, vel = nb.init_state(n, 1234)
forms = (("scalar-load", {"NB_FAST_PAIRS": "0"}), ("LDS wave form", {"NB_FAST_PAIRS": "0", "NB_FAST_SL": "0"}), ("pairs form", {"NB_FAST_PAIRS": "1"}))
for force3d in ("0", "1"):
    ref = None
    for name, env in forms:
        for k in ("NB_FAST_SL", "NB_FAST_PAIRS"):
            os.environ.pop(k, None)
        os.environ.update(env)
        os.environ["NB_FORCE_3D"] = force3d
        nb.reload_env()
        with nb.Scene(pos, vel, nb.default_params(mode=nb.NB_MODE_FAST)) as sc:
            sc.step_n(1)
            p1, v1 = sc.state()
            sc.step_n(3)
            sc.sync()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                sc.step_n(steps)
                sc.sync()
                best = min(best, (time.perf_counter() - t0) / steps * 1e3)
        if ref is None:
            ref = v1
        print(f"N={n} force_3d={force3d} {name:16s}: {best:.3f} ms/step   max|dv| vs first after one step {np.abs(v1 - ref).max():.2e}", flush=True)
for k in ("NB_FAST_SL", "NB_FAST_PAIRS", "NB_FORCE_3D"):
    os.environ.pop(k, None)
nb.reload_env()
