"""The host side of the C ABI under AddressSanitizer + UndefinedBehaviorSanitizer ("optional
-fsanitize=address host build of the C ABI").  CPU only -- GPU sanitizer runs are not available on this pool -- and only
the HOST code of nb_api.hip / nb_shard.inc is instrumented (`make -C nenbody_amd/csrc asan`); the device code objects are
the product's.  The sanitized library is loaded into a fresh interpreter (the runtime must come first in the link order:
LD_PRELOAD) and tests/test_abi.py + tests/test_partition.py run against it: argument validation, the no-device paths, the
shard contract, the plan arithmetic over ragged shapes."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "nenbody_amd", "csrc")
ASAN_LIB = os.path.join(ROOT, "nenbody_amd", "lib", "libnenbody_hip_asan.so")


def _hipcc():
    return shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else None)

def test_host_abi_under_address_and_ub_sanitizers():

    import nenbody_amd

    if nenbody_amd.load().nb_device_count() > 0:
        pytest.skip("sanitizer runs are for the GPU-less box only")
    if _hipcc() is None:
        pytest.skip("no hipcc: the sanitizer build cannot be made here")
    subprocess.run(["make", "-s", "-C", CSRC, "asan"], check=True, stdout=subprocess.DEVNULL)
    clang = os.path.join(os.path.dirname(os.path.realpath(_hipcc())), "..", "lib", "llvm", "bin", "clang")
    if not os.path.exists(clang):
        clang = "/opt/rocm/lib/llvm/bin/clang"
    rt = subprocess.run([clang, "--print-file-name=libclang_rt.asan-x86_64.so"], check=True, capture_output=True, text=True).stdout.strip()
    assert os.path.exists(rt), rt
    env = dict(os.environ)
    env.update({"NENBODY_LIB": ASAN_LIB, "LD_PRELOAD": rt, "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0:exitcode=86",
                "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1:exitcode=87"})
    code = ("import sys, pytest, nenbody_amd._lib as L; assert L.LIB_PATH.endswith('_asan.so'), L.LIB_PATH; "
            "maps = open('/proc/self/maps').read(); L.load(); maps = open('/proc/self/maps').read(); "
            "assert 'libnenbody_hip_asan.so' in maps and 'libclang_rt.asan' in maps; "
            "sys.exit(pytest.main(['-x', '-q', '-p', 'no:cacheprovider', 'tests/test_abi.py', 'tests/test_partition.py']))")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, f"sanitized host ABI run failed (exit {r.returncode}; 86 = ASan, 87 = UBSan):\n{tail}"
    assert "passed" in r.stdout and "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail
