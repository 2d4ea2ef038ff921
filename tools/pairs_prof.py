#!/usr/bin/env python3
"""STEPS (default 150) FAST steps at N = 131072 in the pairs form, for rocprofv3 (--kernel-trace --stats: the per-kernel split of a
step; --pmc passes).  Usage: pairs_prof.py [STEPS]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["NB_FAST_PAIRS"] = "1"
import nenbody_amd as nb  # noqa: E402
nb.reload_env()  # tools/ read the NB_* kernel-form knobs; a host that merely loads the library does not (nb_diag_enable_env)

pos, vel = nb.init_state(131072, 1234)
with nb.Scene(pos, vel, nb.default_params(mode=nb.NB_MODE_FAST)) as sc:
    sc.step_n(int(sys.argv[1]) if len(sys.argv) > 1 else 150)
    sc.sync()
