#!/usr/bin/env python3
"""A few boids-controller steps on the whole set (for rocprofv3): boids_run.py [N [REPS]]; NB_BOIDS_* select the form."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nenbody_amd as nb  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
pos, vel = nb.init_state(n, 1234)
with nb.Scene(pos, vel) as sc:
    sc.step_boids_n(reps)
    sc.sync()
print("done", n, reps, {k: v for k, v in os.environ.items() if k.startswith("NB_")})
