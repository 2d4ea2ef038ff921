"""Independent numpy float32 restatement of src/main.rs:404-441, used to cross-check the C oracle.

Vectorised over the bodies n, sequential over the fold index i, so every body sees the reference's
summation order.  numpy float32 arrays do elementwise IEEE binary32 arithmetic with no fusion.
"""
import numpy as np

F = np.float32


def step(pos, vel, dt=F(0.1), g=F(0.001), bias=F(0.0000001)):
    old = pos.astype(np.float32).copy()          # main.rs:415
    vel = vel.astype(np.float32).copy()
    n = len(old)
    s = np.zeros((n, 3), np.float32)             # main.rs:426
    for i in range(n):                           # main.rs:425 fold order
        vec = old[i][None, :] - old              # main.rs:428  p_i - p_n, for all n at once
        d = old[i][None, :] - old                # distance2 = (other - self).magnitude2()
        sq = d * d
        dist = ((sq[:, 0] + sq[:, 1]) + sq[:, 2]) + F(bias)     # main.rs:429
        s = s + (vec * F(g)) / dist[:, None]     # main.rs:430
    vel = vel + s * F(dt)                        # main.rs:434
    new = vel + old                              # main.rs:436
    return new.astype(np.float32), vel.astype(np.float32)


def instances(pos, vel):
    """main.rs:437-439 with main.rs:141-143, closed form (c, s, 0, 0 / -s, c, 0, 0 / 0 0 1 0 / p 1)."""
    n = len(pos)
    theta = np.arctan2(vel[:, 1].astype(np.float32), vel[:, 0].astype(np.float32)).astype(np.float32)
    s, c = np.sin(theta).astype(np.float32), np.cos(theta).astype(np.float32)
    m = np.zeros((n, 4, 4), np.float32)
    m[:, 0, 0], m[:, 0, 1] = c, s
    m[:, 1, 0], m[:, 1, 1] = -s, c
    m[:, 2, 2] = 1
    m[:, 3, :3] = pos
    m[:, 3, 3] = 1
    return m
