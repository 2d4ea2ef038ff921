"""GPU tests of SURVEY.md section 8f rows 2-4: device-resident hand-off of positions / velocities / model matrices,
CameraArray::update (src/gfx.rs:397-408) and the random-walk controller (src/main.rs:381-402)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def perspective_cp(fovy_deg=45.0, aspect=16 / 9, near=1.0, far=10000.0):
    """OPENGL_TO_WGPU_MATRIX * cgmath::perspective(...) (gfx.rs:365-368), column-major, as the caller would hand it over."""
    f = np.float32(1.0 / np.tan(np.deg2rad(fovy_deg) / 2))
    proj = np.zeros((4, 4), np.float32)        # proj[k] = column k
    proj[0, 0] = f / np.float32(aspect)
    proj[1, 1] = f
    proj[2, 2] = np.float32((far + near) / (near - far))
    proj[2, 3] = -1
    proj[3, 2] = np.float32(2 * far * near / (near - far))
    corr = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0.5, 0], [0, 0, 0.5, 1]], np.float32)   # columns
    cp = np.zeros((4, 4), np.float32)
    for k in range(4):
        for r in range(4):
            cp[k, r] = sum(np.float32(corr[c, r]) * np.float32(proj[k, c]) for c in range(4))
    return cp.astype(np.float32)


def test_camera_constant_of_the_library_is_the_callers(nb):
    """nb_camera_constant against the restatement above (numpy's tan in binary64, so values, not bits)"""
    assert np.allclose(nb.camera_constant(45.0, 16 / 9), perspective_cp(), rtol=2e-6, atol=0)


def test_cameras_bit_exact(nb, oracle):
    n = 3000
    pos, vel = oracle.init_state(n, 11)
    pos[:, 2] = np.linspace(-5, 5, n, dtype=np.float32)
    vel[:, 2] = np.float32(0.01)
    up = np.array([0, 0, 1], np.float32)          # the eye cameras' shared normal
    cp = perspective_cp()
    with nb.Scene(pos, vel) as sc:
        got = sc.cameras(up, cp)
        assert (bits(sc.cameras(up, nb.camera_constant(30.0, 1.5))) == bits(oracle.cameras(pos, vel, up, oracle.camera_constant(30.0, 1.5)))).all()
        sc.step_n(2)
        got2 = sc.cameras(up, cp)
        p2, v2 = sc.state()
    ref = oracle.cameras(pos, vel, up, cp)
    assert (bits(got) == bits(ref)).all()
    assert (bits(got2) == bits(oracle.cameras(p2, v2, up, cp))).all()
    # sanity: with cp = identity the matrix is look_at_dir: orthonormal rotation part
    with nb.Scene(pos, vel) as sc:
        view = sc.cameras(up, np.eye(4, dtype=np.float32))
    rot = view[:, :3, :3]
    assert np.allclose(np.einsum("nij,nkj->nik", rot, rot), np.eye(3), atol=1e-5)


def test_cameras_degenerate_directions_like_the_reference(nb, oracle):
    """Zero velocity -> normalize divides by zero; direction parallel to up -> zero cross product: NaNs in the same places."""
    pos = np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9]], np.float32)
    vel = np.array([[0, 0, 0], [0, 0, 2], [1, 0, 0]], np.float32)
    up = np.array([0, 0, 1], np.float32)
    with nb.Scene(pos, vel) as sc:
        got = sc.cameras(up, np.eye(4, dtype=np.float32))
    ref = oracle.cameras(pos, vel, up, np.eye(4, dtype=np.float32))
    assert (np.isnan(got) == np.isnan(ref)).all()
    ok = ~np.isnan(ref)
    assert (bits(got)[ok] == bits(ref)[ok]).all()


def test_random_controller_matches_oracle_stream(nb, oracle):
    n = 5000
    pos, vel = oracle.init_state(n, 3)
    with nb.Scene(pos, vel) as sc:
        sc.step_random(seed=99, k=3)
        p, v = sc._positions.copy(), sc._velocities.copy()
        inst = sc._instances.copy()
        sc.step_n(1)                 # n-body step in between: the step counter keeps running
        sc.step_random(seed=99, k=2)
        p2, v2 = sc._positions.copy(), sc._velocities.copy()
    pr, vr, ir = oracle.random_run(pos, vel, 3, seed=99, want_instances=True)
    assert (bits(p) == bits(pr)).all() and (bits(v) == bits(vr)).all()
    assert np.allclose(inst, ir, rtol=0, atol=1e-6)
    # distribution of the jitter (main.rs:393-394): U[-1e-4, 1e-4) per step on x and y, nothing on z
    dv = (v - vel).astype(np.float64)
    assert (dv[:, 2] == 0).all() and np.abs(dv[:, :2]).max() < 3e-4 and abs(dv[:, :2].mean()) < 5e-6
    assert 0.8e-4 < dv[:, :2].std() < 1.2e-4          # sum of 3 uniforms of half-width 1e-4: sigma = 1e-4
    pr2, vr2 = oracle.run(pr, vr, 1)
    pr2, vr2 = oracle.random_run(pr2, vr2, 2, seed=99, first_step=4)
    assert (bits(p2) == bits(pr2)).all() and (bits(v2) == bits(vr2)).all()


@pytest.mark.parametrize("n", [300, 20000])          # both transfer paths of the drop-in calls (small / large sets)
def test_update_instance_random_operator(nb, oracle, n):
    """The reference's third controller as one call with its own three slices (main.rs:381-385): frame after frame it is the
    oracle's stream (seed, step, n); its zip (main.rs:386-389) bounds which bodies move and the rest are left alone."""
    pos, vel = oracle.init_state(n, 8)
    p, v, inst = pos.copy(), vel.copy(), np.zeros((n, 4, 4), np.float32)
    for step in range(3):
        nb.update_instance_random_seeded(inst, p, v, seed=1234, step=step)
    pr, vr, ir = oracle.random_run(pos, vel, 3, seed=1234, want_instances=True)
    assert (bits(p) == bits(pr)).all() and (bits(v) == bits(vr)).all()
    assert np.allclose(inst, ir, rtol=0, atol=1e-6)
    # the reference's own three-argument signature (main.rs:381-385): seed and frame counter live in the library
    nb.update_random_seed(1234)
    p3, v3, inst3 = pos.copy(), vel.copy(), np.zeros((n, 4, 4), np.float32)
    for _ in range(3):
        nb.update_instance_random(inst3, p3, v3)
    assert (bits(p3) == bits(pr)).all() and (bits(v3) == bits(vr)).all() and (bits(inst3) == bits(inst)).all()
    nb.update_random_seed(1234)    # ... and restarting the stream repeats the run
    p4, v4 = pos.copy(), vel.copy()
    nb.update_instance_random(inst3, p4, v4)
    pr1, vr1 = oracle.random_run(pos, vel, 1, seed=1234)
    assert (bits(p4) == bits(pr1)).all() and (bits(v4) == bits(vr1)).all()
    # a shorter instance slice: only its bodies move; positions and velocities past it keep their bits
    m = n // 3
    p2, v2, inst2 = pos.copy(), vel.copy(), np.zeros((m, 4, 4), np.float32)
    nb.update_instance_random_seeded(inst2, p2, v2, seed=7, step=5)
    pr2, vr2 = oracle.random_run(pos[:m], vel[:m], 1, seed=7, first_step=5)
    assert (bits(p2[:m]) == bits(pr2)).all() and (bits(v2[:m]) == bits(vr2)).all()
    assert (bits(p2[m:]) == bits(pos[m:])).all() and (bits(v2[m:]) == bits(vel[m:])).all()
    nb.update_instance_random(np.zeros((0, 4, 4), np.float32), p2, v2)      # an empty zip moves nothing
    assert (bits(p2[m:]) == bits(pos[m:])).all()
    with pytest.raises(TypeError):
        nb.update_instance_random(inst2, p2.astype(np.float64), v2)
    nb.update_release()


def test_device_state_handoff(nb, oracle):
    """Zero-copy hand-off: the device pointers the context hands out hold the current records and model matrices."""
    import torch

    n = 1000
    pos, vel = oracle.init_state(n, 5)
    hip = ctypes.CDLL(None)        # the one HIP runtime of this process
    with nb.Scene(pos, vel) as sc:
        sc.step_n(2)
        p_dev, v_dev, m_dev = sc.device_state()
        sc.sync()
        assert p_dev and v_dev and m_dev
        rec = np.zeros((n, 4), np.float32)
        vrec = np.zeros((n, 4), np.float32)
        mats = np.zeros((n, 4, 4), np.float32)
        for dst, src in ((rec, p_dev), (vrec, v_dev), (mats, m_dev)):
            assert hip.hipMemcpy(ctypes.c_void_p(dst.ctypes.data), ctypes.c_void_p(src), ctypes.c_size_t(dst.nbytes), 2) == 0
        p, v = sc.state()
        inst = sc.instances()
    assert (bits(rec[:, :3]) == bits(p)).all() and (rec[:, 3] == 0).all()
    assert (bits(vrec[:, :3]) == bits(v)).all()
    assert (bits(mats) == bits(inst)).all()
    p_ref, v_ref = oracle.run(pos, vel, 2)
    assert (bits(p) == bits(p_ref)).all()
    del torch


def test_model_matrices_are_bit_identical(nb, oracle, monkeypatch):
    """Row a7 (src/main.rs:437-439, rotation_of :141-143) word for word: M = T(p) * Rz(atan2(v.y, v.x)) with the angle, its sine
    and its cosine computed on the device as the HOST's libm computes them (nenbody_amd/csrc/nb_libm.h: glibc's atan2f / sinf /
    cosf restated; the oracle calls the host libm itself, as the reference's f32::atan2 / sin_cos do).  A million drawn velocities
    -- the step's own range, every binade from 2^-60 to 2^60 in either component, near-axis and near-diagonal directions -- and the
    edge cases: signed zeros, subnormals, the axes, infinities, NaN, x = 1 exactly (atan2f's shortcut), ratios at atanf's interval
    boundaries.  The device's own libm (NB_INST_DEVICE_LIBM=1, the form until round 3) stays within 1e-6 and differs in the last
    place somewhere: the control arm."""
    import torch

    from nenbody_amd.dist import HipBackend

    rng = np.random.default_rng(2024)
    m = 1 << 20
    vel = np.zeros((m, 3), np.float32)
    q = m // 4
    vel[:q, :2] = rng.uniform(-0.2, 0.2, (q, 2))                                          # what a step produces
    e = rng.integers(-60, 61, (q, 2))
    vel[q:2 * q, :2] = (rng.uniform(1, 2, (q, 2)) * np.exp2(e) * rng.choice([-1, 1], (q, 2))).astype(np.float32)   # every binade
    ang = rng.uniform(-np.pi, np.pi, q)
    r = np.exp2(rng.uniform(-20, 20, q))
    vel[2 * q:3 * q, 0], vel[2 * q:3 * q, 1] = (r * np.cos(ang)).astype(np.float32), (r * np.sin(ang)).astype(np.float32)
    k = np.arange(q)                                                                        # near the axes and the diagonals
    base = np.array([[1, 0], [0, 1], [-1, 0], [0, -1], [1, 1], [-1, 1], [1, -1], [-1, -1]], np.float32)[k % 8]
    vel[3 * q:, :2] = base + rng.uniform(-1, 1, (q, 2)).astype(np.float32) * np.exp2(rng.integers(-30, -1, (q, 1))).astype(np.float32)
    sub, big, inf, nan = np.float32(1e-42), np.float32(3e38), np.float32(np.inf), np.float32(np.nan)
    edge = [(0.0, 0.0), (-0.0, 0.0), (0.0, -0.0), (-0.0, -0.0), (1.0, 0.0), (0.0, 1.0), (-1.0, 0.0), (0.0, -1.0), (sub, sub), (-sub, sub),
            (sub, 1.0), (1.0, sub), (big, big), (-big, big), (big, -sub), (inf, 1.0), (1.0, inf), (-inf, inf), (inf, -inf), (nan, 1.0),
            (1.0, nan), (1.0, 0.4375), (1.0, 0.6875), (1.0, 1.1875), (1.0, 2.4375), (1.0, 2.0 ** 25), (1.0, 2.0 ** -29), (2.0, 0.875),
            (-3.0, 7.3125), (1.0, 1.0), (-1.0, -1.0), (0.5, 2.0 ** 61), (-0.5, 2.0 ** -61), (-2.0 ** 61, 0.5)]
    vel[:len(edge), 0] = [a for a, _ in edge]
    vel[:len(edge), 1] = [b for _, b in edge]
    pos = rng.uniform(-100, 100, (m, 3)).astype(np.float32)
    pos[:8] = [[0, 0, 0], [-0.0, -0.0, -0.0], [inf, 1, 2], [nan, 1, 2], [1e-40, -1e-40, 3], [3e38, -3e38, 1], [1, 2, 3], [-1, -2, -3]]
    ref = oracle.instances(pos, vel)
    be, dev = HipBackend(), torch.device("cuda", 0)

    def rec(a):
        t = torch.zeros((m, 4))
        t[:, :3] = torch.from_numpy(a)
        return t.to(dev)

    def device_matrices():
        inst = torch.zeros((m, 16), device=dev)
        be.instances(m, rec(pos), rec(vel), inst)
        torch.cuda.synchronize()
        return inst.cpu().numpy().reshape(m, 4, 4)

    got = device_matrices()
    same = (bits(got) == bits(ref)) | (np.isnan(got) & np.isnan(ref))
    bad = np.argwhere(~same.reshape(m, 16).all(axis=1)).ravel()
    assert len(bad) == 0, f"{len(bad)} of {m} matrices differ; first: velocity {vel[bad[0]]!r}: {got[bad[0]].ravel()!r} vs {ref[bad[0]].ravel()!r}"
    # the context API and the drop-in call produce them through the same device function
    with nb.Scene(pos[:4096], vel[:4096]) as sc:
        assert (((bits(sc.instances()) == bits(ref[:4096])) | np.isnan(ref[:4096])).all())
    # control arm: the device's own atan2f / sinf / cosf
    monkeypatch.setenv("NB_INST_DEVICE_LIBM", "1")
    old = device_matrices()
    finite = np.isfinite(ref).all(axis=(1, 2)) & np.isfinite(old).all(axis=(1, 2))
    assert np.abs(old[finite] - ref[finite])[:, :2, :2].max() <= 1e-6
    differing = int((bits(old[finite]) != bits(ref[finite])).any(axis=(1, 2)).sum())
    print(f"device libm: {differing} of {int(finite.sum())} matrices differ from the host's in some word")
    assert differing > 0


def test_library_selftest_of_the_matrices_against_the_host_libm(nb):
    """nb_selftest_matrices (include/nenbody_diag.h): the check an integrator runs on a new host -- four million seeded velocities
    through the device kernel and through the host's own atan2f / sinf / cosf, no mismatch on the hosts this was built for"""
    bad, where = ctypes.c_uint64(123), (ctypes.c_float * 2)()
    assert nb.load().nb_selftest_matrices(1 << 22, 2024, ctypes.byref(bad), where) == 0, nb._lib.last_error()
    assert bad.value == 0, f"{bad.value} matrices differ from the host libm's, e.g. for velocity ({where[0]!r}, {where[1]!r})"
    assert nb.load().nb_selftest_matrices(0, 1, ctypes.byref(bad), None) == nb._lib.NB_ERR_INVALID


def test_device_libm_equals_the_host_libm_argument_by_argument(nb):
    """nb_selftest_libm: the device's restated sinf / cosf / atanf / atan2f against the host's, over windows of consecutive
    bit patterns that cover every branch -- around 0, the subnormals, 2^-12 and 2^-29 (the "return x" cut-offs), pi/4, 1, the
    reduction switch at 120, 2^25, infinity and NaN, both signs.  (tools/libm_device.py runs all 2^32 arguments of each:
    profiles/r04/libm_device.log.)"""
    lib = nb.load()
    bad, where = ctypes.c_uint64(), ctypes.c_uint32()
    windows = [0x00000000, 0x007ff000, 0x30fff000, 0x397ff000, 0x3ee00000 - 0x800, 0x3f300000 - 0x800, 0x3f490000, 0x3f7ff800,
               0x3f980000 - 0x800, 0x401c0000 - 0x800, 0x42effc00, 0x4bfff800, 0x7f7ff000]
    for fn in (0, 1, 2):
        for w in windows:
            for sign in (0, 0x80000000):
                assert lib.nb_selftest_libm(fn, w | sign, 1 << 13, 0, ctypes.byref(bad), ctypes.byref(where)) == 0
                assert bad.value == 0, f"fn {fn}: {bad.value} mismatches from 0x{(w | sign):08x}, first 0x{where.value:08x}"
        assert lib.nb_selftest_libm(fn, 0x3f000000, 1 << 24, 0, ctypes.byref(bad), ctypes.byref(where)) == 0 and bad.value == 0   # [0.5, 2)
    for xb in (0x3f800000, 0xbf000000, 0x3dcccccd, 0x00000000, 0x7f800000):
        for w in (0x00000000, 0x3d000000, 0x3f000000, 0x42000000, 0x7f7ff000, 0xbd000000, 0xbf000000):
            assert lib.nb_selftest_libm(3, w, 1 << 16, xb, ctypes.byref(bad), ctypes.byref(where)) == 0
            assert bad.value == 0, f"atan2f(y from 0x{w:08x}, x = 0x{xb:08x}): first bad y 0x{where.value:08x}"
    assert lib.nb_selftest_libm(7, 0, 1, 0, ctypes.byref(bad), None) == nb._lib.NB_ERR_INVALID
