#!/usr/bin/env python3
"""Where the drop-in call starts to pay: per-call time of update_instance_nbody / update_instance_boids (upload + one step +
download, STRICT) against the CPU restatement of the same function on this host's cores, over the reference's own sizes
(entity_count = 100, main.rs:654; "TODO: Support entity counts higher than 2048", main.rs:653) and beyond.
The CPU column is the oracle (a C port of the Rust, all host cores and one core): a baseline, not the product."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nenbody_amd as nb  # noqa: E402
nb.reload_env()  # tools/ read the NB_* kernel-form knobs; a host that merely loads the library does not (nb_diag_enable_env)
import oracle  # noqa: E402  (baseline leg only)


def per_call(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


for n in [int(x) for x in sys.argv[1:]] or [100, 512, 2048, 8192, 32768, 131072]:
    pos, vel = nb.init_state(n, 1234)
    inst = np.zeros((n, 4, 4), np.float32)
    p, v = pos.copy(), vel.copy()
    op, ov = np.zeros_like(p), np.zeros_like(v)
    reps = max(3, min(200, int(2e9 / (float(n) * n + 1e6))))
    g_nbody = per_call(lambda: nb.update_instance_nbody(inst, p, op, v, ov), reps)
    p, v = pos.copy(), vel.copy()
    g_boids = per_call(lambda: nb.update_instance_boids(inst, p, op, v, ov), reps)
    creps = max(1, min(50, int(5e8 / (float(n) * n + 1e5))))
    c_all = per_call(lambda: oracle.run(pos, vel, 1), creps)
    c_one = per_call(lambda: oracle.run(pos, vel, 1, threads=1), max(1, creps // 8)) if n <= 32768 else float("nan")
    print(f"N={n:7d}  GPU drop-in n-body {g_nbody * 1e6:10.1f} us  boids {g_boids * 1e6:10.1f} us   |  CPU n-body all {oracle.ncores()} cores "
          f"{c_all * 1e6:12.1f} us, one core {c_one * 1e6:12.1f} us", flush=True)
