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


def boids_step(pos, vel, dt=F(0.04), r1=F(1000.0), r2=F(5.0), r3=F(500.0), s1=F(0.02), s2=F(0.05), s3=F(0.5)):
    """Independent numpy float32 restatement of update_instance_boids, src/main.rs:443-526 (vectorised over the
    bodies n, sequential over the fold index i)."""
    old_p = pos.astype(np.float32).copy()        # main.rs:459
    old_v = vel.astype(np.float32).copy()        # main.rs:460
    n = len(old_p)
    idx = np.arange(n)
    c = np.zeros((n, 3), np.float32)
    r = np.zeros((n, 3), np.float32)
    m = np.zeros((n, 3), np.float32)
    cnt = np.zeros(n, np.int32)
    vcnt = np.zeros(n, np.int32)
    for i in range(n):
        d = old_p[i][None, :] - old_p            # distance2: (other - self)
        sq = d * d
        d2 = (sq[:, 0] + sq[:, 1]) + sq[:, 2]
        ne = idx != i
        p1 = (d2 < F(r1)) & ne                   # main.rs:474-475
        c = np.where(p1[:, None], c + old_p[i][None, :], c)
        cnt = cnt + p1
        p2 = (np.sqrt(d2) < F(r2)) & ne          # main.rs:485-486
        r = np.where(p2[:, None], r - (old_p[i][None, :] - old_p), r)
        dv = old_v[i][None, :] - old_v
        sv = dv * dv
        d2v = (sv[:, 0] + sv[:, 1]) + sv[:, 2]
        p3 = (np.sqrt(d2v) < F(r3)) & ne         # main.rs:497-498
        m = np.where(p3[:, None], m + old_v[i][None, :], m)
        vcnt = vcnt + p3
    has = cnt > 0
    c = np.where(has[:, None], c / np.maximum(cnt, 1).astype(np.float32)[:, None], c)        # main.rs:506-508
    hasv = vcnt > 0
    m = np.where(hasv[:, None], m / np.maximum(vcnt, 1).astype(np.float32)[:, None], m)     # main.rs:510-512
    v = (c * F(s1) + r * F(s2)) + m * F(s3)      # main.rs:514
    sq = v * v
    mag = np.sqrt((sq[:, 0] + sq[:, 1]) + sq[:, 2])
    big = mag > F(1.0)                           # main.rs:516-518
    scale = np.where(big, F(1.0) / np.where(big, mag, F(1.0)), F(1.0)).astype(np.float32)
    v = np.where(big[:, None], v * scale[:, None], v).astype(np.float32)
    p = (v * F(dt) + old_p).astype(np.float32)   # main.rs:521
    return p, v
