"""nenbody_amd/csrc/nb_libm.h -- glibc's atan2f / sinf / cosf restated for the device (row a7: bit-identical model matrices,
src/main.rs:437-439, 141-143) -- against the HOST's libm, on the CPU.  The same header compiles for the device (nb_aux.inc) and
for the host; here tools/libm_exhaustive.c is built with gcc and run over a stride of the 2^32 arguments of sinf, cosf and atanf
and over atan2f's special cases, an exponent grid and drawn pairs.  The full run -- every argument, 4.3e9 pairs: 0 mismatches --
is profiles/r04/libm_exhaustive.log (40 s on 8 cores); `-m gpu` holds the device to the oracle's matrices
(tests/test_gpu_rows_f.py::test_model_matrices_are_bit_identical)."""
import os
import subprocess

from conftest import ROOT


def test_restated_libm_equals_the_host_libm(tmp_path):
    exe = str(tmp_path / "libm_exhaustive")
    r = subprocess.run(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-fopenmp", "-I", os.path.join(ROOT, "nenbody_amd", "csrc"),
                        os.path.join(ROOT, "tools", "libm_exhaustive.c"), "-lm", "-o", exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, "1021", "20000000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.splitlines()
    assert len(lines) == 4 and all(" 0 mismatches" in ln for ln in lines), r.stdout


def test_the_unfused_build_differs_somewhere(tmp_path):
    """control arm: built as glibc's x86-64 BASELINE sinf / cosf would be (no contraction: NB_LIBM_FMA=0), the restatement no longer
    matches this host's libm on every argument -- the contraction read from libm.so.6's disassembly is load-bearing, and the check
    above can fail"""
    import torch  # noqa: F401  (only to skip on hosts without FMA, where glibc selects the baseline build itself)

    if "fma" not in open("/proc/cpuinfo").read():
        import pytest

        pytest.skip("host without FMA: glibc runs the baseline build")
    exe = str(tmp_path / "libm_unfused")
    r = subprocess.run(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-fopenmp", "-DNB_LIBM_FMA=0", "-I", os.path.join(ROOT, "nenbody_amd", "csrc"),
                        os.path.join(ROOT, "tools", "libm_exhaustive.c"), "-lm", "-o", exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, "3", "1000"], capture_output=True, text=True, timeout=900)
    sin_line, cos_line, atan_line = r.stdout.splitlines()[:3]
    assert " 0 mismatches" not in sin_line or " 0 mismatches" not in cos_line, r.stdout
    assert " 0 mismatches" in atan_line   # atanf is binary32 throughout and never contracted
