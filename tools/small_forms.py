#!/usr/bin/env python3
"""The launch forms against each other on SMALL sets (device-resident loops, microseconds per step): where make_plan,
fast_split and boids_use_pc draw their lines.  small_forms.py [strict|fast|boids ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nenbody_amd as nb  # noqa: E402
nb._lib.use_library(nb._lib.LEGACY_LIB_PATH)   # this script names launch shapes only the legacy build holds (make -C nenbody_amd/csrc legacy)
nb.reload_env()  # tools/ read the NB_* kernel-form knobs; a host that merely loads the library does not (nb_diag_enable_env)

KNOBS = ("NB_STRICT_BC", "NB_STRICT_PC", "NB_FAST_WAVES", "NB_FAST_IB", "NB_FAST_SLICES", "NB_BOIDS_PC", "NB_TILE")
what = sys.argv[1:] or ["strict", "fast", "boids"]


def timed(n, env, run, steps=2000):
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update({k: str(v) for k, v in env.items()})
    nb.reload_env()
    try:
        dt = run(steps)
        print(f"n={n:6d} {str(env):60s} {dt * 1e6:8.2f} us/step", flush=True)
    except Exception as e:  # a shape the library refuses at this size
        print(f"n={n:6d} {env} refused: {e}", flush=True)


for n in (100, 256, 512, 1024, 2048, 3072, 4096, 8192):
    pos, vel = nb.init_state(n, 1234)

    def nbody(mode):
        def run(steps):
            with nb.Scene(pos, vel, nb.default_params(mode=mode)) as sc:
                sc.step_n(50)
                sc.sync()
                t0 = time.perf_counter()
                sc.step_n(steps)
                sc.sync()
                return (time.perf_counter() - t0) / steps
        return run

    def boids(steps):
        with nb.Scene(pos, vel) as sc:
            sc.step_boids_n(50)
            sc.sync()
            t0 = time.perf_counter()
            sc.step_boids_n(steps)
            sc.sync()
            return (time.perf_counter() - t0) / steps

    if "strict" in what:
        for env in ({}, {"NB_STRICT_BC": 1}, {"NB_STRICT_BC": 0, "NB_STRICT_PC": 14}, {"NB_STRICT_BC": 0, "NB_STRICT_PC": 0}):
            timed(n, env, nbody(nb.NB_MODE_STRICT))
    if "fast" in what:
        for env in ({}, {"NB_FAST_WAVES": 0}, {"NB_FAST_WAVES": 4}, {"NB_FAST_WAVES": 8, "NB_FAST_IB": 2}, {"NB_FAST_WAVES": 8, "NB_FAST_SLICES": 1},
                    {"NB_FAST_WAVES": 8, "NB_FAST_SLICES": 2}, {"NB_FAST_WAVES": 8, "NB_FAST_SLICES": 4}):
            timed(n, env, nbody(nb.NB_MODE_FAST))
    if "boids" in what:
        for env in ({}, {"NB_BOIDS_PC": 3}, {"NB_BOIDS_PC": 2}, {"NB_BOIDS_PC": 1}):
            timed(n, env, boids, steps=1000)
for k in KNOBS:
    os.environ.pop(k, None)
