#!/usr/bin/env python3
"""A few boids-controller steps (for rocprofv3): boids_run.py [N [REPS [COUNT]]] -- the whole set through the context API, or
with COUNT one rank's share [0, COUNT) of the N-body set through the launch API; NB_BOIDS_* select the form."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nenbody_amd as nb  # noqa: E402
nb.reload_env()  # tools/ read the NB_* kernel-form knobs; a host that merely loads the library does not (nb_diag_enable_env)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
count = int(sys.argv[3]) if len(sys.argv) > 3 else 0
pos, vel = nb.init_state(n, 1234)
if count:
    import torch

    from nenbody_amd.dist import HipBackend

    be, dev = HipBackend(), torch.device("cuda", 0)

    def rec(a):
        t = torch.zeros((n, 4))
        t[:, :3] = torch.from_numpy(a)
        return t.to(dev)

    pin, vin = rec(pos), rec(vel)
    pout, vout = torch.zeros_like(pin), torch.zeros_like(vin)
    for _ in range(reps):
        be.boids_step(nb.default_boids_params(), n, 0, count, pin, vin, pout, vout)
    torch.cuda.synchronize()
else:
    with nb.Scene(pos, vel) as sc:
        sc.step_boids_n(reps)
        sc.sync()
print("done", n, reps, count, {k: v for k, v in os.environ.items() if k.startswith("NB_")})
