"""CPU tests of the oracle's boids controller (update_instance_boids, src/main.rs:443-526): hand-derived known
answers and an independent numpy restatement.  The reference has no tests for it either (parity unpinned)."""
import numpy as np
import pytest

import np_restatement as npr

F = np.float32


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_boids_constants_are_the_references(oracle):
    bp = oracle.boids_params()   # src/main.rs:450-456
    assert (bp.dt, bp.rule_1_distance, bp.rule_2_distance, bp.rule_3_distance) == (F(0.04), F(1000.0), F(5.0), F(500.0))
    assert (bp.rule_1_scale, bp.rule_2_scale, bp.rule_3_scale) == (F(0.02), F(0.05), F(0.5))


def test_boids_single_body_stands_still(oracle):
    """n == i is excluded from every fold, so a lone boid gets velocity 0 and stays put (main.rs:475,486,498,514)."""
    p, v = oracle.boids_run(np.array([[3, 4, 5]], np.float32), np.array([[0.5, 0.5, 0]], np.float32), 3, threads=1)
    assert (v == 0).all() and (p == np.array([[3, 4, 5]], np.float32)).all()


def test_boids_two_body_known_answer(oracle):
    """Bodies at (0,0,0) and (3,0,0), velocities (0.1,0,0) and (0,0.2,0).  Hand evaluation of main.rs:471-521 for body 0:
    distance2 = 9 < 1000 -> center = (3,0,0)/1; distance = 3 < 5 -> repel = 0 - (3,0,0) = (-3,0,0);
    velocity distance = sqrt(0.01+0.04) < 500 -> match = (0,0.2,0)/1.
    vel = (3*0.02 + -3*0.05) + 0*0.5 = 0.06 - 0.15 in x, (0 + 0) + 0.2*0.5 in y; |vel| < 1, no clamp; pos = vel*0.04 + pos."""
    pos = np.array([[0, 0, 0], [3, 0, 0]], np.float32)
    vel = np.array([[0.1, 0, 0], [0, 0.2, 0]], np.float32)
    p, v = oracle.boids_run(pos, vel, 1, threads=1)
    vx = (F(3) * F(0.02) + F(-3) * F(0.05)) + F(0) * F(0.5)
    vy = (F(0) * F(0.02) + F(-0.0) * F(0.05)) + F(0.2) * F(0.5)
    assert v[0, 0] == vx and v[0, 1] == vy and v[0, 2] == 0
    assert p[0, 0] == vx * F(0.04) + F(0) and p[0, 1] == vy * F(0.04) + F(0)
    # body 1 sees body 0: center (0,0,0), repel = -(0-3) = +3 in x, match (0.1,0,0)
    assert v[1, 0] == (F(0) * F(0.02) + F(3) * F(0.05)) + F(0.1) * F(0.5)


def test_boids_radius_predicates(oracle):
    """rule 1 compares the SQUARED distance with 1000 (main.rs:474-475); rule 2 the distance with 5 (main.rs:485-486)."""
    pos = np.array([[0, 0, 0], [31, 0, 0], [32, 0, 0], [4.9, 0, 0], [0, 5.0, 0]], np.float32)
    vel = np.zeros((5, 3), np.float32)
    _, v = oracle.boids_run(pos, vel, 1, threads=1)
    # body 0: rule 1 neighbours: 31 (961 < 1000), 4.9, (0,5); not 32 (1024).  center = ((31+4.9+0)/3, 5/3, 0)
    cx = ((F(31) + F(4.9)) + F(0)) / F(3)
    cy = ((F(0) + F(0)) + F(5)) / F(3)
    # rule 2 neighbours: only 4.9 (distance 5.0 is not < 5): repel = -(4.9, 0, 0)
    vx = (cx * F(0.02) + (F(0) - F(4.9)) * F(0.05)) + F(0) * F(0.5)
    vy = (cy * F(0.02) + F(0) * F(0.05)) + F(0) * F(0.5)
    assert v[0, 0] == vx and v[0, 1] == vy


def test_boids_speed_clamp(oracle):
    """|vel| > 1 is rescaled by 1/|vel| (normalize_to(1.0), main.rs:516-518)."""
    pos = np.array([[100, 100, 0], [110, 100, 0], [100, 110, 0]], np.float32)
    vel = np.zeros((3, 3), np.float32)
    p, v = oracle.boids_run(pos, vel, 1, threads=1)
    speed = np.sqrt((v.astype(np.float64) ** 2).sum(axis=1))
    assert np.allclose(speed, 1.0, atol=1e-6)    # center*0.02 has magnitude ~3 -> clamped
    assert np.allclose(p - pos, v * F(0.04), atol=2e-5)   # positions near 100: one ulp is 7.6e-6


@pytest.mark.parametrize("n,k", [(1, 2), (2, 3), (33, 3), (200, 3), (700, 2)])
def test_boids_matches_independent_numpy_restatement_bit_for_bit(oracle, n, k):
    pos, vel = oracle.init_state(n, seed=500 + n)
    pos[:, 2] = np.linspace(-20, 20, n, dtype=np.float32)
    vel[:, 2] = F(0.03)
    pos *= F(0.3)                 # tighter cloud: more bodies inside the rule-2 radius
    po, vo = oracle.boids_run(pos, vel, k)
    pn, vn = pos, vel
    for _ in range(k):
        pn, vn = npr.boids_step(pn, vn)
    assert (bits(po) == bits(pn)).all()
    assert (bits(vo) == bits(vn)).all()


def test_boids_thread_count_and_range_invariance(oracle):
    pos, vel = oracle.init_state(300, seed=77)
    ref = oracle.boids_run(pos, vel, 3, threads=1)
    got = oracle.boids_run(pos, vel, 3, threads=5)
    assert (bits(ref[0]) == bits(got[0])).all() and (bits(ref[1]) == bits(got[1])).all()
    p1, v1 = oracle.boids_run(pos, vel, 1, threads=1)
    for first, count in [(0, 300), (10, 50), (299, 1)]:
        p, v = oracle.boids_step_range(pos, vel, first, count)
        assert (bits(p) == bits(p1[first:first + count])).all() and (bits(v) == bits(v1[first:first + count])).all()
