#!/usr/bin/env python3
"""Stress: every STRICT launch form must produce the same bits, step after step (a synchronisation bug in the block-chain or
producer/consumer kernels would show as a mismatch or as NaN).  Runs shards of several sizes and offsets through the launch
API with each form and compares positions and velocities bit for bit against one lane per body.
`stress_forms.py ROUNDS boids` does the same for the boids controller's five launch forms (one lane per body plain / packed,
producer/consumer, chain split plain / packed), with data that mixes planar and 3-D tiles, velocities on both sides of the
rule-3 bound and radii that cut."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import nenbody_amd as nb  # noqa: E402
nb._lib.use_library(nb._lib.LEGACY_LIB_PATH)   # this script names launch shapes only the legacy build holds (make -C nenbody_amd/csrc legacy)
nb.reload_env()  # tools/ read the NB_* kernel-form knobs; a host that merely loads the library does not (nb_diag_enable_env)
from nenbody_amd.dist import HipBackend  # noqa: E402

FORMS = {"lane": {"NB_STRICT_PC": "0", "NB_STRICT_LANES": "1", "NB_STRICT_BC": "0"}, "bc": {"NB_STRICT_BC": "1"},
         "pc14": {"NB_STRICT_PC": "14", "NB_STRICT_BC": "0"}}
be = HipBackend()
dev = torch.device("cuda", 0)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
bad = 0
t0 = time.perf_counter()


def boids_stress():
    global bad
    forms = {"lane": "3", "packed": "2", "pc": "1", "split": "4", "split-packed": "5"}
    for rnd in range(rounds):
        rng = np.random.default_rng(7000 + rnd)
        for n, shards, steps in ((131072, [(0, 16384), (49152, 65536)], 2), (50001, [(0, 20000), (20000, 30001)], 3), (6000, [(0, 6000)], 8),
                                 (70000, [(1, 69999)], 2), (4097, [(1, 4096)], 6)):
            pos, vel = nb.init_state(n, 2000 + rnd)
            pos *= np.float32(rng.choice([0.05, 0.3, 1.0]))
            kind = rnd % 4
            bp = nb.default_boids_params()
            if kind == 1:      # some 3-D tiles
                lo = int(rng.integers(0, n - 600))
                pos[lo:lo + 500, 2] = rng.uniform(-5, 5, 500).astype(np.float32)
                vel[lo + 100:lo + 300, 2] = np.float32(0.03)
            elif kind == 2:    # a rule-3 radius that cuts, velocities on both sides of the bound in different tiles
                bp.rule_3_distance = 0.2
                vel[::3] *= np.float32(4.0)
                lo = int(rng.integers(0, n - 3000))
                vel[lo:lo + 2048] *= np.float32(0.1)
            elif kind == 3:    # other radii, a non-finite record
                bp.rule_1_distance, bp.rule_2_distance = 40.0, 9.0
                pos[int(rng.integers(0, n)), 0] = np.inf
            def rec(a):
                t = torch.zeros((n, 4)); t[:, :3] = torch.from_numpy(a); return t.to(dev)
            p0, v0 = rec(pos), rec(vel)
            results = {}
            for name, knob in forms.items():
                os.environ["NB_BOIDS_PC"] = knob
                nb.reload_env()
                pc, vc = p0.clone(), v0.clone()
                pn, vn = p0.clone(), v0.clone()
                for _ in range(steps):
                    for first, count in shards:
                        be.boids_step(bp, n, first, count, pc, vc, pn, vn)
                    torch.cuda.synchronize()
                    mask = torch.ones(n, dtype=torch.bool, device=dev)
                    for first, count in shards:
                        mask[first:first + count] = False
                    pn[mask] = pc[mask]
                    vn[mask] = vc[mask]
                    pc, pn = pn, pc
                    vc, vn = vn, vc
                results[name] = (pc.cpu().numpy().view(np.uint32), vc.cpu().numpy().view(np.uint32))
            ref = results["lane"]
            for name in forms:
                if name == "lane":
                    continue
                same = (results[name][0] == ref[0]).all() and (results[name][1] == ref[1]).all()
                bad += 0 if same else 1
                print(f"boids round {rnd} kind={kind} n={n} shards={shards} steps={steps} {name}: {'same bits' if same else 'MISMATCH'}", flush=True)
    os.environ.pop("NB_BOIDS_PC", None)
    nb.reload_env()


if len(sys.argv) > 2 and sys.argv[2] == "boids":
    boids_stress()
    print(f"{'OK' if bad == 0 else 'FAILED'}: {bad} mismatching runs, {time.perf_counter() - t0:.0f} s")
    sys.exit(1 if bad else 0)
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
