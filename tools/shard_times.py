#!/usr/bin/env python3
"""One rank's share of a step, timed with events: shard_times.py [strict|fast|boids] -- shards of 16 384 / 32 768 / 65 536 / all of
131 072 bodies (what a rank of an 8 / 4 / 2 / 1-GPU job launches per step), the library's own kernel form for each, ms per step
and the compute-only speed-up over the whole set.  3-D data with NB_FORCE_3D=1."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import nenbody_amd as nb  # noqa: E402
nb.reload_env()  # tools/ read the NB_* kernel-form knobs; a host that merely loads the library does not (nb_diag_enable_env)
from nenbody_amd.dist import HipBackend  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "strict"   # strict | fast | boids | boids-split
n_total = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
be = HipBackend()
dev = torch.device("cuda", 0)
pos, vel = nb.init_state(n_total, 1234)
cur = torch.zeros((n_total, 4)); cur[:, :3] = torch.from_numpy(pos); cur = cur.to(dev)
nxt = torch.zeros_like(cur)
vin = torch.zeros((n_total, 4)); vin[:, :3] = torch.from_numpy(vel); vin = vin.to(dev)
vout = torch.zeros_like(vin)
params = nb.default_params(mode=nb.NB_MODE_FAST if what == "fast" else nb.NB_MODE_STRICT)
bp = nb.default_boids_params()
res = {}
for count in (n_total, n_total // 2, n_total // 4, n_total // 8):
    v4 = torch.zeros((count, 4), device=dev)
    sb = be.scratch_bytes(params, n_total, count)
    scratch = torch.empty((sb,), dtype=torch.uint8, device=dev) if sb else None

    bscratch = torch.empty((max(16, be.boids_split_scratch_bytes(bp, n_total, count)),), dtype=torch.uint8, device=dev)

    def step():
        if what == "boids-split":   # the j range in slices: the reference's neighbour sets and counts, reassociated sums
            be.boids_step_split(bp, n_total, 0, count, cur, vin, nxt, vout, bscratch)
        elif what == "boids":
            be.boids_step(bp, n_total, 0, count, cur, vin, nxt, vout)
        else:
            be.step(params, n_total, 0, count, cur, nxt, v4, scratch)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            step()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    res[count] = best
    kern = nb._lib.planned_kernels(params, n_total, count)[0] if not what.startswith("boids") else what
    print(f"{what} N={n_total} shard {count:7d}: {best:.3f} ms/step  x{res[n_total] / best:.2f} of the whole set  [{kern}]", flush=True)
