#!/usr/bin/env python3
"""Soak run of the STRICT division ladder against the IEEE divide (nb_selftest_divide): ladder_soak.py [ROUNDS [LOG2_PAIRS]].
Each round draws 2^LOG2_PAIRS (numerator, denominator) pairs over the guarded exponent rectangle of the reference constants
with a fresh seed; any mismatch is printed with an offending pair.  One line per round (a long run must keep talking)."""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nenbody_amd as nb  # noqa: E402
nb.reload_env()  # tools/ read the NB_* kernel-form knobs; a host that merely loads the library does not (nb_diag_enable_env)
from nenbody_amd import _lib  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
log2 = int(sys.argv[2]) if len(sys.argv) > 2 else 40
lib = _lib.load()
p = nb.default_params()
total_bad = 0
t_start = time.perf_counter()
for r in range(rounds):
    bad = ctypes.c_uint64(0)
    pair = np.zeros(2, np.float32)
    t0 = time.perf_counter()
    rc = lib.nb_selftest_divide(ctypes.byref(p), 1 << log2, 0x5eed0000 + r, ctypes.byref(bad), pair.ctypes.data)
    if rc != 0:
        raise SystemExit(f"nb_selftest_divide failed: {_lib.last_error()}")
    total_bad += bad.value
    print(f"round {r}: 2^{log2} pairs, seed {0x5eed0000 + r:#x}, {bad.value} mismatches"
          + (f" (e.g. n={pair[0]!r} d={pair[1]!r})" if bad.value else "") + f", {time.perf_counter() - t0:.1f} s", flush=True)
print(f"total: {rounds} x 2^{log2} = {rounds * (1 << log2):.3e} pairs, {total_bad} mismatches, {time.perf_counter() - t_start:.0f} s")
