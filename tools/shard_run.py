#!/usr/bin/env python3
"""Run one rank's share of a STRICT/FAST step a few times (for rocprofv3): shard_run.py COUNT [REPS] (env selects the kernel shape)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import nenbody_amd as nb  # noqa: E402
nb.reload_env()  # tools/ read the NB_* kernel-form knobs; a host that merely loads the library does not (nb_diag_enable_env)
from nenbody_amd.dist import HipBackend  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mode = nb.NB_MODE_FAST if os.environ.get("NB_MODE") == "fast" else nb.NB_MODE_STRICT
n_total = 131072
be = HipBackend()
dev = torch.device("cuda", 0)
pos, vel = nb.init_state(n_total, 1234)
cur = torch.zeros((n_total, 4)); cur[:, :3] = torch.from_numpy(pos); cur = cur.to(dev)
nxt = torch.zeros_like(cur)
params = nb.default_params(mode=mode)
v4 = torch.zeros((count, 4), device=dev)
sb = be.scratch_bytes(params, n_total, count)
scratch = torch.empty((sb,), dtype=torch.uint8, device=dev) if sb else None
if os.environ.get("NB_MODE") == "boids":
    vin = torch.zeros((n_total, 4)); vin[:, :3] = torch.from_numpy(vel); vin = vin.to(dev)
    vout = torch.zeros_like(vin)
    bp = nb.default_boids_params()
    for _ in range(reps):
        be.boids_step(bp, n_total, 0, count, cur, vin, nxt, vout)
else:
    for _ in range(reps):
        be.step(params, n_total, 0, count, cur, nxt, v4, scratch)
torch.cuda.synchronize()
print("done", count, reps)
