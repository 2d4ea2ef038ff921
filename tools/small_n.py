#!/usr/bin/env python3
"""Per-step wall time of device-resident step_n(K) on SMALL sets (launch-bound territory): small_n.py [N ...]
Under `rocprofv3 --kernel-trace --stats` the kernel's own average duration shows how much of the step is gap."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nenbody_amd as nb  # noqa: E402
nb.reload_env()  # tools/ read the NB_* kernel-form knobs; a host that merely loads the library does not (nb_diag_enable_env)

sizes = [int(x) for x in sys.argv[1:]] or [100, 256, 1024, 2048, 4096, 16384]
for n in sizes:
    pos, vel = nb.init_state(n, 1234)
    for mode, name in ((nb.NB_MODE_STRICT, "strict"), (nb.NB_MODE_FAST, "fast")):
        with nb.Scene(pos, vel, nb.default_params(mode=mode)) as sc:
            k = 2000 if n <= 4096 else 500
            sc.step_n(50)
            sc.sync()
            t0 = time.perf_counter()
            sc.step_n(k)
            sc.sync()
            dt = (time.perf_counter() - t0) / k
        print(f"n={n:6d} {name:6s} {dt * 1e6:8.2f} us/step  {1 / dt:10.0f} steps/s  {n / dt:.3e} body-updates/s", flush=True)
    with nb.Scene(pos, vel) as sc:
        k = 2000 if n <= 4096 else 200
        sc.step_boids_n(20)
        sc.sync()
        t0 = time.perf_counter()
        sc.step_boids_n(k)
        sc.sync()
        dt = (time.perf_counter() - t0) / k
    print(f"n={n:6d} boids  {dt * 1e6:8.2f} us/step  {1 / dt:10.0f} steps/s  {n / dt:.3e} body-updates/s", flush=True)
