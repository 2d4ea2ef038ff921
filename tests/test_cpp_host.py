"""The C++ host mirror (include/nenbody_scene.hpp): compiles against the C ABI with plain g++ (CPU test), fails
loudly without a GPU, and on a GPU gives the oracle's bits (GPU test)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

EXE = os.path.join(ROOT, "build", "scene_check")


def build_exe():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    libdir = os.path.join(ROOT, "nenbody_amd", "lib")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "scene_check.cpp"), "-o", EXE, "-L", libdir, "-lnenbody_hip",
                    f"-Wl,-rpath,{libdir}"], check=True)


def test_cpp_host_compiles_and_refuses_to_run_without_a_gpu(nb, tmp_path):
    build_exe()
    from nenbody_amd import _lib

    if _lib.load().nb_device_count() > 0:
        pytest.skip("a HIP device is present")
    r = subprocess.run([EXE, "64", "2", str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 10 and "no HIP device" in r.stderr     # NB_ERR_NO_DEVICE surfaced as nenbody::Error


@pytest.mark.gpu
@pytest.mark.parametrize("n,k", [(300, 3), (4096, 2)])
def test_cpp_host_matches_oracle(nb, oracle, tmp_path, n, k):
    build_exe()
    out = tmp_path / "out.bin"
    r = subprocess.run([EXE, str(n), str(k), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = np.fromfile(out, dtype=np.float32)
    p, v, inst = raw[:3 * n].reshape(n, 3), raw[3 * n:6 * n].reshape(n, 3), raw[6 * n:].reshape(n, 4, 4)
    pos, vel = oracle.init_state(n, 1234)
    p_ref, v_ref, inst_ref = oracle.run(pos, vel, k + 1, want_instances=True)   # scene_check takes k steps + 1
    assert (p.view(np.uint32) == p_ref.view(np.uint32)).all()
    assert (v.view(np.uint32) == v_ref.view(np.uint32)).all()
    assert np.allclose(inst, inst_ref, rtol=0, atol=1e-6)
