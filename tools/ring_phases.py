#!/usr/bin/env python3
"""What the step in PHASES costs (nb_launch_ring_fold_phase), compute only, on one GPU:

    python -u tools/ring_phases.py [N_TOTAL [WORLDS [C4_OWN_LIST [C4_REST_LIST]]]]      e.g.  131072 2,4,8 0,4,8 0,32,40

For the first and the last rank of every world: the one-launch fold + finish (nb_launch_ring_fold), against the same step as a
host with the exchanges hidden issues it (OWN, REST, SUMS, finish; and OWN_READY, REST, the fused finish), and the phases one by
one -- OWN is what the all-gather can hide behind, SUMS what the second exchange can.
C4 lists: sub-tiles per workgroup of the own-slot / rest phases to try (0 = the library's choice); NB_RING_CAP from the environment.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import nenbody_amd as nb  # noqa: E402
from nenbody_amd.dist import HipBackend  # noqa: E402

n_total = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
worlds = [int(w) for w in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2, 4, 8]
c4_own_list = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
c4_rest_list = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0]
L = nb._lib
be = HipBackend()
dev = torch.device("cuda", 0)
pos, vel = nb.init_state(n_total, 1234)
cur = torch.zeros((n_total, 4)); cur[:, :3] = torch.from_numpy(pos); cur = cur.to(dev)
nxt = torch.zeros_like(cur)
params = nb.default_params(mode=nb.NB_MODE_FAST)


def timed(step, reps=20):
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            step()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best * 1e3   # microseconds


# preheat: the whole set on one GPU
v4 = torch.zeros((n_total, 4), device=dev)
scratch = torch.empty((be.scratch_bytes(params, n_total, n_total),), dtype=torch.uint8, device=dev)
whole = timed(lambda: be.step(params, n_total, 0, n_total, cur, nxt, v4, scratch), 30)
whole = timed(lambda: be.step(params, n_total, 0, n_total, cur, nxt, v4, scratch), 30)
print(f"N={n_total} whole set, one GPU: {whole:.1f} us/step", flush=True)
del scratch
for world in worlds:
    S = n_total // world
    v4 = torch.zeros((S, 4), device=dev)
    for c4o in c4_own_list:
        for c4r in c4_rest_list:
            for k, v in (("NB_RING_C4_OWN", c4o), ("NB_RING_C4_REST", c4r)):
                if v:
                    os.environ[k] = str(v)
                else:
                    os.environ.pop(k, None)
            nb.reload_env()
            D = be.ring_partners(params, n_total, 0, S)
            if D == 0 or not be.ring_phased(params, n_total, 0, S):
                print(f"  {world} ranks: shard {S}: no phases for this shape")
                continue
            sums = torch.zeros(((D + 1) * S, 4), device=dev)
            recv = torch.zeros((D * S, 4), device=dev)
            for r in (0, world - 1):
                a = (params, n_total, r * S, S)
                scratch = torch.empty((be.ring_scratch_bytes(*a),), dtype=torch.uint8, device=dev)   # (the rows' layout depends on the rank's lists)

                def ph(p):
                    be.ring_fold_phase(*a, p, cur, sums, scratch)

                def fin():
                    be.ring_finish(*a, cur, nxt, v4, sums, recv)

                def one():
                    be.ring_fold(*a, cur, sums, scratch)
                    fin()

                def phased():
                    ph(L.NB_RING_OWN); ph(L.NB_RING_REST); ph(L.NB_RING_SUMS); fin()

                # the fused finish: no SUMS launch, and from the second step on no planes launch in front of OWN (the positions
                # alternate between two replicas as a host's do: each step starts on what the finish before wrote)
                pp = [cur.clone(), nxt.clone()]

                def fused(first=False):
                    be.ring_fold_phase(*a, L.NB_RING_OWN if first else L.NB_RING_OWN_READY, pp[0], sums, scratch)
                    be.ring_fold_phase(*a, L.NB_RING_REST, pp[0], sums, scratch)
                    be.ring_finish_phase(*a, pp[0], pp[1], v4, None, recv, scratch)
                    pp.reverse()

                t1, tp = timed(one), timed(phased)
                fused(first=True)
                tf = timed(fused)
                parts = {name: timed(lambda p=p: ph(p)) for name, p in (("OWN", L.NB_RING_OWN), ("REST", L.NB_RING_REST), ("SUMS", L.NB_RING_SUMS))}
                print(f"  {world} ranks, rank {r}, shard {S} (D={D}), c4 own/rest {c4o or 'dflt'}/{c4r or 'dflt'} cap {os.environ.get('NB_RING_CAP', 'dflt')}: "
                      f"one launch {t1:.1f} us x{whole / t1:.2f} | in phases {tp:.1f} ({tp - t1:+.1f}) | with the fused finish {tf:.1f} ({tf - t1:+.1f}) | "
                      + " ".join(f"{k} {v:.1f}" for k, v in parts.items()) + f" | finish {timed(fin):.1f}", flush=True)
            del sums, recv
