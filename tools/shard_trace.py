#!/usr/bin/env python3
"""A native shard's steps for a kernel trace: rank 0 of WORLD (default 8) of N = 131 072 in the pairs form, every exchange on a
one-rank RCCL communicator.  shard_trace.py [phases|sequence] [STEPS]   -- run under rocprofv3 --kernel-trace and read the gaps."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nenbody_amd as nb  # noqa: E402

nb.reload_env()
mode = sys.argv[1] if len(sys.argv) > 1 else "phases"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
n, world = 131072, 8
pos, vel = nb.init_state(n, 1234)
nb.load().nb_diag_rccl_solo(1)
sh = nb.NativeShard(pos, vel, nb.default_params(mode=nb.NB_MODE_FAST), rank=0, world=world, comm_id=nb.comm_id(), overlap=mode == "phases", pairs=True)
assert sh.partners and sh.pairs_overlapped == (mode == "phases")
sh.step(20)
sh.sync()
sh.step(steps)
sh.sync()
sh.close()
print("done", mode, steps)
