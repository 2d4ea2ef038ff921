#!/usr/bin/env python3
"""One FAST whole-set step a few times (for rocprofv3): fast_run.py [N [REPS]]; the NB_FAST_* environment selects the form."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nenbody_amd as nb  # noqa: E402
nb.reload_env()  # tools/ read the NB_* kernel-form knobs; a host that merely loads the library does not (nb_diag_enable_env)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
pos, vel = nb.init_state(n, 1234)
with nb.Scene(pos, vel, nb.default_params(mode=nb.NB_MODE_FAST)) as sc:
    sc.step_n(reps)
    sc.sync()
print("done", n, reps, {k: v for k, v in os.environ.items() if k.startswith("NB_")})
