#!/usr/bin/env python3
"""One FAST step of a large set in the pairs form (default 4 194 304 bodies: 32 chunks, 528 tiles), a few bodies against the
binary64 sum beside the reference's own binary32 fold, and the step's wall time.  Usage: pairs_big.py [N]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nenbody_amd as nb  # noqa: E402
nb.reload_env()  # tools/ read the NB_* kernel-form knobs; a host that merely loads the library does not (nb_diag_enable_env)
import oracle  # noqa: E402  (the checker)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
pos, vel = nb.init_state(n, 4321)
fast = nb.default_params(mode=nb.NB_MODE_FAST)
print(f"N={n}: {nb._lib.planned_kernels(fast, n, n)}", flush=True)
with nb.Scene(pos, vel, fast) as sc:
    t0 = time.perf_counter()
    sc.step_n(1)
    sc.sync()
    t1 = time.perf_counter()
    sc.step_n(1)
    sc.sync()
    t2 = time.perf_counter()
with nb.Scene(pos, vel, fast) as sc:
    sc.step_n(1)
    _, v1 = sc.state()
# against the binary64 sum: FAST must be no further from it than the reference's own binary32 fold is (at this size both lose
# digits to cancellation: the net force on a body in a uniform square is a small difference of large sums)
c = [float(np.float32(x)) for x in (0.1, 0.001, 0.0000001)]
idx = np.unique(np.concatenate([[0, n - 1, 131071, 131072], np.linspace(0, n - 1, 6).astype(np.int64)]))
refs = np.concatenate([oracle.step_range(pos, vel[i:i + 1], int(i), 1)[1] for i in idx])
dv64 = np.concatenate([oracle.step_range_dv_f64(pos, int(i), 1, *c) for i in idx])
v_true = vel[idx].astype(np.float64) + dv64
scale = np.abs(dv64).max()
err_ref = np.abs(refs.astype(np.float64) - v_true).max(axis=1)
err_fast = np.abs(v1[idx].astype(np.float64) - v_true).max(axis=1)
print(f"N={n}: first step {1e3 * (t1 - t0):.1f} ms, second {1e3 * (t2 - t1):.1f} ms; max |v - v64| / max|dv| on {len(idx)} bodies: reference binary32 "
      f"{err_ref.max() / scale:.2e}, FAST pairs form {err_fast.max() / scale:.2e}", flush=True)
assert (err_fast <= err_ref + 2e-5 * scale + float(np.spacing(np.float32(np.abs(v_true).max())))).all()
