#!/usr/bin/env python3
"""nb_selftest_libm over whole functions: the device's sinf / cosf / atanf (nenbody_amd/csrc/nb_libm.h) against the host's libm on
every one of the 2^32 binary32 arguments, and atan2f(y, x) over all y for a few x -- libm_device.py [sin cos atan atan2]."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nenbody_amd as nb  # noqa: E402
nb.reload_env()  # tools/ read the NB_* kernel-form knobs; a host that merely loads the library does not (nb_diag_enable_env)

lib = nb.load()
which = sys.argv[1:] or ["sin", "cos", "atan", "atan2"]
fns = {"sin": 0, "cos": 1, "atan": 2}
bad, where = ctypes.c_uint64(), ctypes.c_uint32()
for name in which:
    if name in fns:
        t0 = time.time()
        nb._lib.check(lib.nb_selftest_libm(fns[name], 0, 0, 0, ctypes.byref(bad), ctypes.byref(where)))
        print(f"{name}f: 4294967296 arguments, {bad.value} mismatches (first at 0x{where.value:08x}), {time.time() - t0:.0f} s", flush=True)
    else:
        for xb in (0x3f800000, 0xbf800000, 0x3dcccccd, 0xbe99999a, 0x00000000, 0x80000000, 0x7f800000, 0x42f60000):
            t0 = time.time()
            nb._lib.check(lib.nb_selftest_libm(3, 0, 0, xb, ctypes.byref(bad), ctypes.byref(where)))
            print(f"atan2f(y, x = 0x{xb:08x}): 4294967296 arguments y, {bad.value} mismatches (first at 0x{where.value:08x}), {time.time() - t0:.0f} s", flush=True)
