#!/usr/bin/env python3
"""STRICT whole-set launch forms at N bodies (default 131072), ms per step: the LDS-tiled kernel (NB_STRICT_SL=0) against the
scalar-load kernel's launch shapes (NB_STRICT_SL = 1: four waves per workgroup, 16 records per planar request; 2: four waves,
8 records; 3: one wave per workgroup, 8 records), planar data and NB_FORCE_3D=1; every form is checked against the first one
bit for bit.  Usage: sl_shapes.py [N [STEPS]]"""
import os
import sys
import time


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nenbody_amd as nb  # noqa: E402
nb._lib.use_library(nb._lib.LEGACY_LIB_PATH)   # this script names launch shapes only the legacy build holds (make -C nenbody_amd/csrc legacy)
nb.reload_env()  # tools/ read the NB_* kernel-form knobs; a host that merely loads the library does not (nb_diag_enable_env)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
pos, vel = nb.init_state(n, 1234)
for force3d in ("0", "1"):
    ref = None
    for sl in ("0", "1", "2", "3"):
        os.environ["NB_STRICT_SL"] = sl
        os.environ["NB_FORCE_3D"] = force3d
        nb.reload_env()
        with nb.Scene(pos, vel) as sc:
            sc.step_n(3)
            sc.sync()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                sc.step_n(steps)
                sc.sync()
                best = min(best, (time.perf_counter() - t0) / steps * 1e3)
            p, v = sc.state()
        same = "" if ref is None else ("  == form 0" if (p.tobytes(), v.tobytes()) == ref else "  DIFFERS from form 0")
        if ref is None:
            ref = (p.tobytes(), v.tobytes())
        print(f"N={n} force_3d={force3d} NB_STRICT_SL={sl}: {best:.3f} ms/step{same}", flush=True)
os.environ.pop("NB_STRICT_SL")
os.environ.pop("NB_FORCE_3D")
nb.reload_env()
