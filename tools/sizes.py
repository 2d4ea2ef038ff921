#!/usr/bin/env python3
"""STRICT / FAST / boids at the BASELINE sizes, device-resident, ms per step and the derived rates:
sizes.py [N ...] (default 1024 16384 131072 1048576)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nenbody_amd as nb  # noqa: E402
nb.reload_env()  # tools/ read the NB_* kernel-form knobs; a host that merely loads the library does not (nb_diag_enable_env)

sizes = [int(a) for a in sys.argv[1:]] or [1024, 16384, 131072, 1 << 20]
for n in sizes:
    pos, vel = nb.init_state(n, 1234)
    steps = max(3, min(200, int(2e11 / (float(n) * n))))
    for name, mode in (("STRICT", nb.NB_MODE_STRICT), ("FAST", nb.NB_MODE_FAST), ("boids", None)):
        if name == "boids" and n >= (1 << 24):
            continue
        params = nb.default_params(mode=mode) if mode is not None else nb.default_params()
        with nb.Scene(pos, vel, params) as sc:
            step = (lambda k: sc.step_boids_n(k)) if name == "boids" else (lambda k: sc.step_n(k))
            step(2)
            sc.sync()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                step(steps)
                sc.sync()
                best = min(best, (time.perf_counter() - t0) / steps)
        kern = nb._lib.planned_kernels(params, n, n)[0] if mode is not None else "boids"
        print(f"N={n:8d} {name:6s}: {best * 1e3:9.4f} ms/step  {1 / best:9.1f} steps/s  {n / best:.3e} body-updates/s  "
              f"{float(n) * n / best:.3e} interactions/s  {18.0 * n * n / best / 1e12:6.1f} TFLOP/s@18  "
              f"{18.0 * n * n / best / 157.3e12:.3f} of fp32 peak  [{kern}]", flush=True)
