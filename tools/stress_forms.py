#!/usr/bin/env python3
"""Stress: every STRICT launch form must produce the same bits, step after step (a synchronisation bug in the block-chain or
producer/consumer kernels would show as a mismatch or as NaN).  Runs shards of several sizes and offsets through the launch
API with each form and compares positions and velocities bit for bit against one lane per body."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import nenbody_amd as nb  # noqa: E402
from nenbody_amd.dist import HipBackend  # noqa: E402

FORMS = {"lane": {"NB_STRICT_PC": "0", "NB_STRICT_LANES": "1", "NB_STRICT_BC": "0"}, "bc": {"NB_STRICT_BC": "1"},
         "pc14": {"NB_STRICT_PC": "14", "NB_STRICT_BC": "0"}}
be = HipBackend()
dev = torch.device("cuda", 0)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
bad = 0
t0 = time.perf_counter()
for rnd in range(rounds):
    for n, shards, steps, three_d in ((131072, [(0, 16384), (16384, 16384), (114688, 16384)], 6, False), (50001, [(0, 20000), (20000, 30001)], 5, True),
                                      (6000, [(0, 6000)], 20, False), (262144, [(65536, 32768)], 3, rnd % 2 == 1), (4097, [(1, 4096)], 10, True)):
        pos, vel = nb.init_state(n, 1000 + rnd)
        if three_d:
            rng = np.random.default_rng(rnd)
            pos[:, 2] = rng.uniform(-50, 50, n).astype(np.float32)
        base = torch.zeros((n, 4)); base[:, :3] = torch.from_numpy(pos); base = base.to(dev)
        results = {}
        for name, env in FORMS.items():
            for k in ("NB_STRICT_PC", "NB_STRICT_LANES", "NB_STRICT_BC"):
                os.environ.pop(k, None)
            os.environ.update(env)
            nb.reload_env()
            params = nb.default_params()
            cur, nxt = base.clone(), base.clone()
            vels = []
            for first, count in shards:
                v = torch.zeros((count, 4)); v[:, :3] = torch.from_numpy(vel[first:first + count]); vels.append(v.to(dev))
            for _ in range(steps):
                for (first, count), v in zip(shards, vels):
                    sb = be.scratch_bytes(params, n, count)
                    scratch = torch.empty((sb,), dtype=torch.uint8, device=dev) if sb else None
                    be.step(params, n, first, count, cur, nxt, v, scratch)
                torch.cuda.synchronize()
                # bodies outside the shards keep their old positions
                mask = torch.ones(n, dtype=torch.bool, device=dev)
                for first, count in shards:
                    mask[first:first + count] = False
                nxt[mask] = cur[mask]
                cur, nxt = nxt, cur
            results[name] = (cur.cpu().numpy().view(np.uint32), [v.cpu().numpy().view(np.uint32) for v in vels])
        ref = results["lane"]
        for name in ("bc", "pc14"):
            same = (results[name][0] == ref[0]).all() and all((a == b).all() for a, b in zip(results[name][1], ref[1]))
            if not same:
                bad += 1
            print(f"round {rnd} n={n} shards={shards} steps={steps} 3d={three_d} {name}: {'same bits' if same else 'MISMATCH'}", flush=True)
print(f"{'OK' if bad == 0 else 'FAILED'}: {bad} mismatching runs, {time.perf_counter() - t0:.0f} s")
sys.exit(1 if bad else 0)
