"""CPU tests of the oracle's restatements of CameraArray::update (src/gfx.rs:397-408) and update_instance_random
(src/main.rs:381-402): hand-derived known answers (the reference has no tests for them either)."""
import numpy as np

F = np.float32


def test_look_at_dir_known_answer(oracle):
    """eye (1,2,3) looking along +x with up = +z: f = (1,0,0); s = f x up = (0*1-0*0, 0*0-1*1, 0) = (0,-1,0);
    u = s x f = (0,0,1).  Columns: (s.x,u.x,-f.x,0) = (0,0,-1,0), (s.y,u.y,-f.y,0) = (-1,0,0,0), (s.z,u.z,-f.z,0) = (0,1,0,0),
    (-eye.s, -eye.u, eye.f, 1) = (2,-3,1,1)."""
    m = oracle.cameras(np.array([[1, 2, 3]], F), np.array([[5, 0, 0]], F), [0, 0, 1], np.eye(4, dtype=F))[0]
    assert (m[0] == np.array([0, 0, -1, 0], F)).all()
    assert (m[1] == np.array([-1, 0, 0, 0], F)).all()
    assert (m[2] == np.array([0, 1, 0, 0], F)).all()
    assert (m[3] == np.array([2, -3, 1, 1], F)).all()


def test_camera_applies_the_constant_on_the_left(oracle):
    """out = cp * view: scaling cp's rows scales the result's rows (gfx.rs:368 associates to the left)."""
    eye, d = np.array([[1, 2, 3]], F), np.array([[0.3, -0.4, 0.1]], F)
    view = oracle.cameras(eye, d, [0, 0, 1], np.eye(4, dtype=F))[0]
    cp = np.diag(np.array([2, 3, 0.5, 1], F))           # column k = k-th unit vector scaled
    out = oracle.cameras(eye, d, [0, 0, 1], cp)[0]
    assert np.allclose(out, view * np.array([2, 3, 0.5, 1], F)[None, :], rtol=1e-6)


def test_random_walk_distribution_and_update_order(oracle):
    n = 20000
    pos = np.zeros((n, 3), F)
    vel = np.zeros((n, 3), F)
    p, v = oracle.random_run(pos, vel, 1, seed=5)
    assert (v[:, 2] == 0).all() and (p[:, 2] == 0).all()                     # main.rs:395
    assert (v[:, :2] >= F(-0.0001)).all() and (v[:, :2] < F(0.0001)).all()     # main.rs:393-394
    assert abs(float(v[:, :2].mean())) < 2e-6 and abs(float(v[:, :2].std()) - 1e-4 / np.sqrt(3)) < 2e-6
    assert (p == v).all()                                                      # pos += the NEW vel, main.rs:397
    # counter based: the same (seed, step, body) gives the same draw, another step or seed a different one
    p2, v2 = oracle.random_run(pos, vel, 1, seed=5)
    assert (v2 == v).all()
    _, v3 = oracle.random_run(pos, vel, 1, seed=5, first_step=1)
    _, v4 = oracle.random_run(pos, vel, 1, seed=6)
    assert not (v3 == v).all() and not (v4 == v).all()
