#!/usr/bin/env python3
"""One rank's share of a FAST step in the pairs form on shards (nb_launch_ring_fold + nb_launch_ring_finish), timed with events:
ring_times.py [n_total] -- ranks of 2 / 4 / 8 (a lower and an upper rank: the antipodal block goes to the lower half of the
ring), compute only (no exchange), against the one-GPU pairs form and against the ordered fold of the same shard.
Kernel shape from the environment: NB_RING_NP, NB_RING_GA, NB_RING_WPB."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import nenbody_amd as nb  # noqa: E402
nb.reload_env()  # tools/ read the NB_* kernel-form knobs; a host that merely loads the library does not (nb_diag_enable_env)
from nenbody_amd.dist import HipBackend  # noqa: E402

n_total = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
worlds = [int(w) for w in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2, 4, 8]
be = HipBackend()
dev = torch.device("cuda", 0)
pos, vel = nb.init_state(n_total, 1234)
cur = torch.zeros((n_total, 4)); cur[:, :3] = torch.from_numpy(pos); cur = cur.to(dev)
nxt = torch.zeros_like(cur)
params = nb.default_params(mode=nb.NB_MODE_FAST)


def timed(step, reps=10):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            step()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


# preheat + the whole set on one GPU
v4 = torch.zeros((n_total, 4), device=dev)
sb = be.scratch_bytes(params, n_total, n_total)
scratch = torch.empty((sb,), dtype=torch.uint8, device=dev)
whole = timed(lambda: be.step(params, n_total, 0, n_total, cur, nxt, v4, scratch), 30 if n_total <= 131072 else 3)
whole = timed(lambda: be.step(params, n_total, 0, n_total, cur, nxt, v4, scratch), 30 if n_total <= 131072 else 3)
print(f"N={n_total} whole set, one GPU [{nb._lib.planned_kernels(params, n_total, n_total)[0]}]: {whole:.3f} ms/step "
      f"(NB_RING_NP={os.environ.get('NB_RING_NP', '-')} GA={os.environ.get('NB_RING_GA', '-')} WPB={os.environ.get('NB_RING_WPB', '-')})", flush=True)
del scratch
for world in worlds:
    S = n_total // world
    v4 = torch.zeros((S, 4), device=dev)
    sb = be.scratch_bytes(params, n_total, S)
    scratch = torch.empty((sb,), dtype=torch.uint8, device=dev)
    ordered = timed(lambda: be.step(params, n_total, 0, S, cur, nxt, v4, scratch), 10 if n_total <= 131072 else 2)
    del scratch
    D = be.ring_partners(params, n_total, 0, S)
    if D == 0:
        print(f"  {world} ranks: shard {S}: ordered fold {ordered:.3f} ms; the shape does not take the ring form")
        continue
    sums = torch.zeros(((D + 1) * S, 4), device=dev)
    recv = torch.zeros((D * S, 4), device=dev)
    scratch = torch.empty((be.ring_scratch_bytes(params, n_total, 0, S),), dtype=torch.uint8, device=dev)
    line = f"  {world} ranks: shard {S}: ordered fold {ordered:.3f} ms x{whole / ordered:.2f};  pairs form (D={D}, scratch {scratch.numel() / 1e6:.0f} MB)"
    for r in (0, world - 1):
        def step():
            be.ring_fold(params, n_total, r * S, S, cur, sums, scratch)
            be.ring_finish(params, n_total, r * S, S, cur, nxt, v4, sums, recv)

        t = timed(step, 10 if n_total <= 131072 else 2)
        line += f"  rank {r}: {t:.3f} ms x{whole / t:.2f}"
    print(line, flush=True)
    del scratch, sums, recv
