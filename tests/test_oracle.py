"""CPU tests of the oracle: hand-derived known answers, an independent numpy restatement, golden vectors.

The reference ships no tests for this path, so these are what pins the oracle.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
import np_restatement as npr

F = np.float32


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_constants_are_the_references(oracle):
    # src/main.rs:411-413
    assert oracle.DT == F(0.1) and oracle.G == F(0.001) and oracle.BIAS == F(0.0000001)


def test_two_body_known_answer(oracle):
    """Bodies at (-1,0,0) and (+1,0,0), at rest.  Hand evaluation of main.rs:428-436 for body 0:
    self term: vec = 0 -> 0*G/bias = 0.  Other: vec = (2,0,0); dist = (4+0)+0 + 1e-7f = 4.0f exactly
    (half an ulp of 4 is 2.4e-7 > 1e-7); term = (2*0.001f)/4.0f; v = 0 + term*0.1f; p = v + (-1)."""
    pos = np.array([[-1, 0, 0], [1, 0, 0]], np.float32)
    vel = np.zeros((2, 3), np.float32)
    p, v = oracle.run(pos, vel, 1, threads=1)
    assert F(4.0) + F(0.0000001) == F(4.0)
    term = (F(2.0) * F(0.001)) / F(4.0)
    vx = F(0.0) + term * F(0.1)
    assert v[0, 0] == vx and v[1, 0] == -vx
    assert p[0, 0] == vx + F(-1.0) and p[1, 0] == -vx + F(1.0)
    assert (v[:, 1:] == 0).all() and (p[:, 1:] == 0).all()


def test_coincident_bodies_contribute_exactly_zero(oracle):
    pos = np.array([[3, 4, 5], [3, 4, 5], [3, 4, 5]], np.float32)
    vel = np.array([[0.5, 0, 0], [0, 0.25, 0], [0, 0, 0.125]], np.float32)
    p, v = oracle.run(pos, vel, 1, threads=1)
    assert (v == vel).all()                     # 0*G/bias == 0 for every pair
    assert (p == vel + pos).all()               # p = v + p, no dt (main.rs:436)


def test_position_update_has_no_dt(oracle):
    pos = np.array([[0, 0, 0]], np.float32)
    vel = np.array([[1, 2, 3]], np.float32)
    p, v = oracle.run(pos, vel, 3, threads=1)
    assert (v == vel).all() and (p == 3 * vel).all()


def test_symmetric_square_is_antisymmetric(oracle):
    pos = np.array([[1, 1, 0], [-1, 1, 0], [-1, -1, 0], [1, -1, 0]], np.float32)
    vel = np.zeros((4, 3), np.float32)
    p, v = oracle.run(pos, vel, 1, threads=1)
    # every body is pulled towards the centre with the same magnitude in x and y
    assert np.allclose(np.abs(v[:, :2]), np.abs(v[0, 0]), rtol=1e-6, atol=0)
    assert (np.sign(v[:, 0]) == -np.sign(pos[:, 0])).all() and (np.sign(v[:, 1]) == -np.sign(pos[:, 1])).all()
    assert abs(float(p[:, 0].sum())) < 1e-6 and abs(float(p[:, 1].sum())) < 1e-6


@pytest.mark.parametrize("n,k", [(1, 2), (2, 3), (17, 3), (64, 3), (257, 2), (1024, 2)])
def test_matches_independent_numpy_restatement_bit_for_bit(oracle, n, k):
    pos, vel = oracle.init_state(n, seed=99 + n)
    pos[:, 2] = np.linspace(-3, 3, n, dtype=np.float32)   # exercise the z component too
    vel[:, 2] = F(0.01)
    po, vo = oracle.run(pos, vel, k)
    pn, vn = pos, vel
    for _ in range(k):
        pn, vn = npr.step(pn, vn)
    assert (bits(po) == bits(pn)).all()
    assert (bits(vo) == bits(vn)).all()


def test_thread_count_never_changes_bits(oracle):
    pos, vel = oracle.init_state(301, seed=5)
    ref = oracle.run(pos, vel, 4, threads=1)
    for t in (2, 3, 8):
        got = oracle.run(pos, vel, 4, threads=t)
        assert (bits(ref[0]) == bits(got[0])).all() and (bits(ref[1]) == bits(got[1])).all()


def test_step_range_equals_full_step(oracle):
    """Sharding by index range cannot change any body's result (the multi-GPU partition relies on this)."""
    pos, vel = oracle.init_state(200, seed=8)
    p_full, v_full = oracle.run(pos, vel, 1, threads=1)
    for first, count in [(0, 200), (0, 7), (7, 100), (107, 93), (199, 1)]:
        p, v = oracle.step_range(pos, vel[first:first + count], first, count)
        assert (bits(p) == bits(p_full[first:first + count])).all()
        assert (bits(v) == bits(v_full[first:first + count])).all()


def test_instances_layout_and_values(oracle):
    pos, vel = oracle.init_state(50, seed=3)
    inst = oracle.instances(pos, vel)
    ref = npr.instances(pos, vel)
    assert np.allclose(inst, ref, rtol=0, atol=1e-6)
    # column-major: column 3 is the translation (consumed as mat4 model[] by shaders/scene.vert:12-14,18)
    assert (inst[:, 3, :3] == pos).all() and (inst[:, 3, 3] == 1).all()
    assert (inst[:, 2] == np.array([0, 0, 1, 0], np.float32)).all()
    # rotation columns are orthonormal, heading along the velocity
    assert np.allclose(inst[:, 0, 0] ** 2 + inst[:, 0, 1] ** 2, 1, atol=1e-6)
    speed = np.linalg.norm(vel[:, :2], axis=1)
    assert np.allclose(inst[:, 0, 0], vel[:, 0] / speed, atol=1e-6)
    assert np.allclose(inst[:, 0, 1], vel[:, 1] / speed, atol=1e-6)


def test_run_returns_instances_of_the_last_step(oracle):
    pos, vel = oracle.init_state(20, seed=4)
    p, v, inst = oracle.run(pos, vel, 3, want_instances=True)
    assert (bits(inst) == bits(oracle.instances(p, v))).all()


def test_init_state_distributions_and_draw_order(oracle):
    n = 4096
    pos, vel = oracle.init_state(n, seed=1234)
    assert (pos[:, 2] == 0).all() and (vel[:, 2] == 0).all()                     # main.rs:740, 745
    assert (vel[:, :2] >= 0).all() and (vel[:, :2] < 0.1).all()                    # U[0, 0.1)
    assert (pos[:, :2] >= -100).all() and (pos[:, :2] < 100).all()                 # U[-100, 100)
    assert abs(float(pos[:, :2].mean())) < 3 and abs(float(vel[:, :2].mean()) - 0.05) < 0.003
    # draw order: all velocities first (main.rs:738-742), then positions (743-747): a shorter set's
    # velocities are a prefix of a longer set's, its positions are not
    pos2, vel2 = oracle.init_state(n // 2, seed=1234)
    assert (vel2 == vel[: n // 2]).all()
    assert not (pos2 == pos[: n // 2]).all()
    # determinism and seed sensitivity
    pos3, vel3 = oracle.init_state(n, seed=1234)
    assert (pos3 == pos).all() and (vel3 == vel).all()
    pos4, _ = oracle.init_state(n, seed=1235)
    assert not (pos4 == pos).all()


def test_golden_vectors(oracle):
    g = np.load(os.path.join(GOLDEN_DIR, "nbody_golden.npz"))
    seed = int(g["seed"][0])
    p0, v0 = oracle.init_state(16, seed)
    assert (bits(p0) == bits(g["n16_init_pos"])).all() and (bits(v0) == bits(g["n16_init_vel"])).all()
    for k in (1, 10):
        p, v = oracle.run(p0, v0, k)
        assert (bits(p) == bits(g[f"n16_k{k}_pos"])).all() and (bits(v) == bits(g[f"n16_k{k}_vel"])).all()
    p0, v0 = oracle.init_state(1024, seed)
    for k in (1, 10, 100):
        p, v = oracle.run(p0, v0, k)
        assert (bits(p) == bits(g[f"n1024_k{k}_pos"])).all() and (bits(v) == bits(g[f"n1024_k{k}_vel"])).all()


def test_golden_1000_steps(oracle):
    """BASELINE config 1: N=1 024, 1 000 steps on the CPU path."""
    g = np.load(os.path.join(GOLDEN_DIR, "nbody_golden.npz"))
    p0, v0 = oracle.init_state(1024, int(g["seed"][0]))
    p, v, inst = oracle.run(p0, v0, 1000, want_instances=True)
    assert (bits(p) == bits(g["n1024_k1000_pos"])).all() and (bits(v) == bits(g["n1024_k1000_vel"])).all()
    assert (bits(inst) == bits(g["n1024_k1000_inst"])).all()


def test_f64_variant_tracks_f32_over_a_short_horizon(oracle):
    """The binary64 recurrence is the noise-floor instrument, not the reference arithmetic."""
    pos, vel = oracle.init_state(256, seed=2)
    p32, _ = oracle.run(pos, vel, 10)
    p64, _ = oracle.run_f64(pos, vel, 10)
    assert np.abs(p32 - p64).max() < 1e-3


@pytest.mark.parametrize("n,k,threads", [(1, 2, 1), (7, 3, 1), (8, 2, 1), (13, 4, 2), (1023, 5, 3), (2051, 2, 2)])
def test_batched_step_is_the_scalar_loop_bit_for_bit(oracle, n, k, threads):
    """nbo_step_range_batched (eight bodies per AVX2 vector; what make_golden.py --c3 steps the headline size with) against
    the scalar restatement of main.rs:424-436: planar and 3-D data, ragged counts, coincident bodies, matrices included."""
    pos, vel = oracle.init_state(n, 77)
    rng = np.random.default_rng(n)
    if n > 8:
        pos[:, 2] = rng.uniform(-50, 50, n).astype(np.float32)
        vel[:, 2] = rng.uniform(0, 0.1, n).astype(np.float32)
        pos[n // 2] = pos[n // 3]                       # a coincident pair: 0 * G / bias
    a = oracle.run(pos, vel, k, threads=threads, want_instances=True)
    b = oracle.run(pos, vel, k, threads=threads, want_instances=True, batched=True)
    for x, y in zip(a, b):
        assert (bits(x) == bits(y)).all()
    if not oracle.load().nbo_batched_available():
        pytest.skip("no AVX2 on this CPU: the batched entry point ran the scalar loop")
