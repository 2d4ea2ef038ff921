#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in forms (diagnostic): Scene.step() refreshes the host mirrors every
step (88 B/body over PCIe); update_instance_nbody() / update_instance_boids() (one nb_update_instance_* call each) additionally
upload the state every call; the device context is kept by the library between calls."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nenbody_amd as nb  # noqa: E402
nb.reload_env()  # tools/ read the NB_* kernel-form knobs; a host that merely loads the library does not (nb_diag_enable_env)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
pos, vel = nb.init_state(n, 1234)
for mode, name in ((nb.NB_MODE_STRICT, "strict"), (nb.NB_MODE_FAST, "fast")):
    with nb.Scene(pos, vel, nb.default_params(mode=mode)) as sc:
        sc.step()
        t0 = time.perf_counter()
        k = 10
        for _ in range(k):
            sc.step()
        dt = (time.perf_counter() - t0) / k
        sc.step_n(3); sc.sync()
        t0 = time.perf_counter()
        sc.step_n(k); sc.sync()
        dr = (time.perf_counter() - t0) / k
    print(f"{name}: Scene.step() incl. download of pos+vel+instances: {dt * 1e3:.3f} ms/step = {n / dt:.3e} body-updates/s; "
          f"device-resident step_n: {dr * 1e3:.3f} ms/step")
inst = np.zeros((n, 4, 4), np.float32)
p, v = pos.copy(), vel.copy()
op, ov = np.zeros_like(p), np.zeros_like(v)
nb.update_instance_nbody(inst, p, op, v, ov)
t0 = time.perf_counter()
for _ in range(5):
    nb.update_instance_nbody(inst, p, op, v, ov)
dt = (time.perf_counter() - t0) / 5
print(f"update_instance_nbody (upload + step + download per call, STRICT): {dt * 1e3:.3f} ms/call = {n / dt:.3e} body-updates/s")
nb.update_instance_boids(inst, p, op, v, ov)
t0 = time.perf_counter()
for _ in range(5):
    nb.update_instance_boids(inst, p, op, v, ov)
dt = (time.perf_counter() - t0) / 5
print(f"update_instance_boids (upload + step + download per call): {dt * 1e3:.3f} ms/call = {n / dt:.3e} body-updates/s")
