#!/usr/bin/env python3
"""Per-call time of the drop-in functions on small sets with the pinned buffer staged by DMA copies (NB_DROPIN_ZERO_COPY=0)
against the pack / unpack kernels touching it through the bus themselves (=1): where kZeroCopyMax belongs; and (round 4) with the
host waiting on the stream (NB_DROPIN_POLL=0) against polling the word the export kernel writes behind its results (the default
up to 2 048 bodies)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nenbody_amd as nb  # noqa: E402
nb.reload_env()  # tools/ read the NB_* kernel-form knobs; a host that merely loads the library does not (nb_diag_enable_env)


def per_call(fn, reps):
    for _ in range(20):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


for n in [int(x) for x in sys.argv[1:]] or [100, 256, 512, 1024, 2048, 4096, 8192, 16384]:
    pos, vel = nb.init_state(n, 1234)
    row = []
    for knob, poll in (("0", "0"), ("1", "0"), ("1", "1")):
        os.environ["NB_DROPIN_ZERO_COPY"] = knob
        os.environ["NB_DROPIN_POLL"] = poll
        nb.reload_env()
        inst = np.zeros((n, 4, 4), np.float32)
        p, v = pos.copy(), vel.copy()
        op, ov = np.zeros_like(p), np.zeros_like(v)
        g_nbody = per_call(lambda: nb.update_instance_nbody(inst, p, op, v, ov), 400)
        p, v = pos.copy(), vel.copy()
        g_boids = per_call(lambda: nb.update_instance_boids(inst, p, op, v, ov), 400)
        with nb.Scene(pos, vel) as sc:
            g_scene = per_call(sc.step, 400)   # one step + download of positions, velocities and matrices
        row.append(f"zero_copy={knob} poll={poll}: n-body {g_nbody * 1e6:7.1f} us boids {g_boids * 1e6:7.1f} us Scene.step {g_scene * 1e6:7.1f} us")
    print(f"N={n:6d}  " + "  |  ".join(row), flush=True)
os.environ.pop("NB_DROPIN_ZERO_COPY", None)
os.environ.pop("NB_DROPIN_POLL", None)
