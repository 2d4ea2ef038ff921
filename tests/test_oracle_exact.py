"""The oracle against IEEE 754 binary32 arithmetic done BY HAND: exact rational arithmetic (fractions.Fraction) with an explicit
round-to-nearest-even into binary32 after every single operation of src/main.rs:425-436 -- no floating-point unit, no compiler
flag, no numpy in the loop.  What this pins: that the C oracle (gcc -O2 -ffp-contract=off on this host) and the numpy restatement
really compute "every operation rounded once, in the reference's order" -- no contraction into FMAs, no excess precision, no
reassociation -- including subnormal results and the coincident / self pair (0 * G / bias = 0).  What it cannot pin is the Rust
original itself (the reference cannot be built here: DESIGN.md section 2, "parity unpinned")."""
from fractions import Fraction

import numpy as np
import pytest


def rnd(x: Fraction) -> Fraction:
    """round a rational to the nearest binary32 (ties to even), subnormals and overflow to infinity handled; exact"""
    if x == 0:
        return Fraction(0)
    s, a = (-1 if x < 0 else 1), abs(x)
    e = a.numerator.bit_length() - a.denominator.bit_length()      # floor(log2 a) or one more
    if Fraction(2) ** e > a:
        e -= 1
    e = max(e, -126)                                                # subnormals share the exponent of the smallest normal
    q = Fraction(2) ** (e - 23)                                     # the spacing of binary32 numbers at this magnitude
    k, r = divmod(a, q)
    k = int(k)
    if r * 2 > q or (r * 2 == q and k % 2 == 1):
        k += 1
    v = k * q
    assert v < Fraction(2) ** 128, "overflow is not exercised by these tests"
    return s * v


def to_frac(f) -> Fraction:
    return Fraction(float(np.float32(f)))


def to_f32(x: Fraction) -> np.float32:
    return np.float32(float(x))        # exact: x is a binary32 value, and binary32 -> binary64 -> binary32 is the identity


def step_exact(pos, vel, dt, g, bias):
    """one application of main.rs:415-436 on lists of Fraction triples, every operation rounded to binary32"""
    old = [tuple(p) for p in pos]
    new_p, new_v = [], []
    for n, (pn, vn) in enumerate(zip(old, vel)):
        acc = [Fraction(0)] * 3                                     # main.rs:426
        for pi in old:                                              # main.rs:425: every body, in index order, i == n included
            vec = [rnd(a - b) for a, b in zip(pi, pn)]              # main.rs:428  p_i - p_n
            d = vec                                                 # distance2: (other - self), the same differences
            sq = [rnd(c * c) for c in d]
            dist = rnd(rnd(rnd(sq[0] + sq[1]) + sq[2]) + bias)      # main.rs:429  ((xx + yy) + zz) + bias
            term = [rnd(rnd(c * g) / dist) for c in vec]            # main.rs:430  (vec * G) / dist, component by component
            acc = [rnd(a + t) for a, t in zip(acc, term)]
        v = [rnd(a + rnd(b * dt)) for a, b in zip(vn, acc)]         # main.rs:434  v + a * dt
        p = [rnd(a + b) for a, b in zip(v, pn)]                     # main.rs:436  v + p  (no dt)
        new_v.append(v)
        new_p.append(p)
    return new_p, new_v


def run_exact(pos, vel, k, dt=0.1, g=0.001, bias=0.0000001):
    dtf, gf, bf = to_frac(dt), to_frac(g), to_frac(bias)
    p = [[to_frac(c) for c in row] for row in pos]
    v = [[to_frac(c) for c in row] for row in vel]
    for _ in range(k):
        p, v = step_exact(p, v, dtf, gf, bf)
    return (np.array([[to_f32(c) for c in row] for row in p], np.float32), np.array([[to_f32(c) for c in row] for row in v], np.float32))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_the_rounding_function_is_binary32():
    rng = np.random.default_rng(1)
    a = rng.standard_normal(2000).astype(np.float32) * np.float32(10.0) ** rng.integers(-30, 30, 2000).astype(np.float32)
    b = rng.standard_normal(2000).astype(np.float32) * np.float32(10.0) ** rng.integers(-12, 12, 2000).astype(np.float32)
    with np.errstate(all="ignore"):
        for x, y in zip(a, b):
            fx, fy = to_frac(x), to_frac(y)
            for exact, want in ((fx + fy, x + y), (fx * fy, x * y), (fx / fy, x / y), (fx - fy, x - y)):
                if np.isfinite(want):                                # (overflow is not exercised: the step never gets there)
                    got = rnd(exact)
                    assert to_f32(got).view(np.uint32) == np.float32(want).view(np.uint32) or (got == 0 and want == 0)
    tiny = np.float32(1e-45)          # the smallest subnormal: halves round to even
    assert rnd(to_frac(tiny) / 2) == 0 and rnd(to_frac(tiny) * Fraction(3, 2)) == to_frac(tiny) * 2


@pytest.mark.parametrize("n,k,seed", [(2, 3, 0), (9, 2, 1), (16, 1, 2), (24, 2, 3)])
def test_oracle_equals_exact_rational_ieee_arithmetic(oracle, n, k, seed):
    pos, vel = oracle.init_state(n, seed)
    rng = np.random.default_rng(seed)
    pos[:, 2] = rng.uniform(-100, 100, n).astype(np.float32)         # 3-D: the z arithmetic too
    if n >= 9:
        pos[3] = pos[7]                                              # a coincident pair: (0 * G) / bias
        pos[5] *= np.float32(1e-20)                                  # differences that round, squares that go subnormal
        vel[2] *= np.float32(1e-30)
    p_ref, v_ref = run_exact(pos, vel, k)
    p, v = oracle.run(pos, vel, k, threads=1)
    assert (bits(p) == bits(p_ref)).all() and (bits(v) == bits(v_ref)).all()
    pb, vb = oracle.run(pos, vel, k, batched=True) if oracle.load().nbo_batched_available() else (p, v)
    assert (bits(pb) == bits(p_ref)).all() and (bits(vb) == bits(v_ref)).all()


def test_other_constants_and_the_numpy_restatement(oracle):
    import np_restatement

    n, k = 12, 2
    pos, vel = oracle.init_state(n, 7)
    pos *= np.float32(0.01)
    for dt, g, bias in ((0.01, 1.0, 1e-3), (1.0, -0.05, 2.0), (0.1, 1e-6, 1e-7)):
        p_ref, v_ref = run_exact(pos, vel, k, dt, g, bias)
        p, v = oracle.run(pos, vel, k, np.float32(dt), np.float32(g), np.float32(bias), threads=1)
        assert (bits(p) == bits(p_ref)).all() and (bits(v) == bits(v_ref)).all(), (dt, g, bias)
        pn, vn = pos, vel
        for _ in range(k):
            pn, vn = np_restatement.step(pn, vn, np.float32(dt), np.float32(g), np.float32(bias))
        assert (bits(pn) == bits(p_ref)).all() and (bits(vn) == bits(v_ref)).all(), (dt, g, bias)


# ---- the boids controller (src/main.rs:443-526) the same way: exact rationals, every operation rounded to binary32 by hand,
#      the square roots through an integer square root with the rounding decided exactly ------------------------------------
def sqrt_rnd(x: Fraction) -> Fraction:
    """the correctly rounded binary32 square root of a non-negative binary32 value (what f32::sqrt / sqrtf return): an integer
    square root to 80 fractional bits picks the candidate, the midpoints to its two binary32 neighbours -- compared exactly, by
    squaring -- decide"""
    if x == 0:
        return Fraction(0)
    from math import isqrt

    t = 100
    m = x * (Fraction(4) ** t)                          # x = m / 4^t with m an integer (x has at most 149 fractional bits)
    assert m.denominator == 1
    c = rnd(Fraction(isqrt(m.numerator), 2 ** t))       # floor(sqrt(m)) / 2^t, rounded to binary32
    for _ in range(3):
        cf = np.float32(float(c))
        lo = (c + to_frac(np.nextafter(cf, np.float32(0)))) / 2
        hi = (c + to_frac(np.nextafter(cf, np.float32(np.inf)))) / 2
        if lo * lo > x:
            c = to_frac(np.nextafter(cf, np.float32(0)))
        elif hi * hi < x:
            c = to_frac(np.nextafter(cf, np.float32(np.inf)))
        else:
            return c                                    # (sqrt(x) is never exactly a midpoint: its square would need 50 bits)
    raise AssertionError("the candidate did not settle")


def boids_exact(pos, vel, k, dt=0.04, r1=1000.0, r2=5.0, r3=500.0, s1=0.02, s2=0.05, s3=0.5):
    dt, r1, r2, r3, s1, s2, s3 = (to_frac(c) for c in (dt, r1, r2, r3, s1, s2, s3))
    p = [[to_frac(c) for c in row] for row in pos]
    v = [[to_frac(c) for c in row] for row in vel]

    def dist2(a, b):                                                       # a.distance2(b) = (b - a).magnitude2()
        e = [rnd(y - x) for x, y in zip(a, b)]
        sq = [rnd(c * c) for c in e]
        return rnd(rnd(sq[0] + sq[1]) + sq[2])

    for _ in range(k):
        old_p, old_v = [list(r) for r in p], [list(r) for r in v]          # main.rs:459-460
        for n in range(len(p)):
            pn, vn = old_p[n], old_v[n]
            c, cnt = [Fraction(0)] * 3, 0
            for i, pi in enumerate(old_p):                                 # main.rs:471-480
                if dist2(pn, pi) < r1 and n != i:
                    c, cnt = [rnd(a + b) for a, b in zip(c, pi)], cnt + 1
            r = [Fraction(0)] * 3
            for i, pi in enumerate(old_p):                                 # main.rs:482-492
                if sqrt_rnd(dist2(pn, pi)) < r2 and n != i:
                    r = [rnd(a - rnd(b - q)) for a, b, q in zip(r, pi, pn)]
            m, vcnt = [Fraction(0)] * 3, 0
            for i, vi in enumerate(old_v):                                 # main.rs:494-504
                if sqrt_rnd(dist2(vn, vi)) < r3 and n != i:
                    m, vcnt = [rnd(a + b) for a, b in zip(m, vi)], vcnt + 1
            if cnt > 0:
                c = [rnd(a / cnt) for a in c]                              # main.rs:506-508  (count as f32: exact below 2^24)
            if vcnt > 0:
                m = [rnd(a / vcnt) for a in m]
            nv = [rnd(rnd(rnd(a * s1) + rnd(b * s2)) + rnd(d * s3)) for a, b, d in zip(c, r, m)]      # main.rs:514
            mag = sqrt_rnd(rnd(rnd(rnd(nv[0] * nv[0]) + rnd(nv[1] * nv[1])) + rnd(nv[2] * nv[2])))
            if mag > 1:                                                    # main.rs:516-518
                s = rnd(Fraction(1) / mag)
                nv = [rnd(a * s) for a in nv]
            v[n] = nv
            p[n] = [rnd(rnd(a * dt) + b) for a, b in zip(nv, pn)]          # main.rs:521
    return (np.array([[to_f32(c) for c in row] for row in p], np.float32), np.array([[to_f32(c) for c in row] for row in v], np.float32))


def test_the_square_root_is_correctly_rounded():
    rng = np.random.default_rng(3)
    xs = np.abs(rng.standard_normal(3000).astype(np.float32)) * np.float32(10.0) ** rng.integers(-20, 20, 3000).astype(np.float32)
    xs = np.concatenate([xs, np.float32([0, 1, 2, 4, 25, 1e-45, 1.1754944e-38, 3.4e38, 24.999998, 25.000002])])
    for x in xs:
        assert to_f32(sqrt_rnd(to_frac(x))).view(np.uint32) == np.sqrt(np.float32(x)).view(np.uint32), x


@pytest.mark.parametrize("n,k,seed,scale", [(8, 2, 0, 0.2), (14, 2, 1, 0.05), (20, 1, 2, 0.5)])
def test_boids_oracle_equals_exact_rational_ieee_arithmetic(oracle, n, k, seed, scale):
    pos, vel = oracle.init_state(n, seed)
    rng = np.random.default_rng(seed)
    pos[:, 2] = rng.uniform(-100, 100, n).astype(np.float32)
    pos *= np.float32(scale)                      # clouds dense enough that all three radii cut somewhere
    vel[:, 2] = rng.uniform(0, 0.1, n).astype(np.float32)
    vel *= np.float32(40.0 if seed == 1 else 1.0)  # seed 1: velocity differences on both sides of rule 3's 500 ... no: of the clamp
    p_ref, v_ref = boids_exact(pos, vel, k)
    p, v = oracle.boids_run(pos, vel, k)
    assert (bits(p) == bits(p_ref)).all() and (bits(v) == bits(v_ref)).all()
    bp = oracle.boids_params()
    bp.rule_3_distance, bp.rule_2_distance, bp.rule_1_distance = 1.5, 3.0, 400.0      # radii that cut
    p_ref, v_ref = boids_exact(pos, vel, 1, r1=400.0, r2=3.0, r3=1.5)
    p, v = oracle.boids_run(pos, vel, 1, bp)
    assert (bits(p) == bits(p_ref)).all() and (bits(v) == bits(v_ref)).all()


# ---- CameraArray::update (src/gfx.rs:397-408, build_camera :358-369): cp * look_at_dir(eye, dir, up), by hand ------------------
def cameras_exact(eyes, dirs, up, cp):
    up = [to_frac(c) for c in up]
    cpf = [[to_frac(c) for c in col] for col in cp]                       # cpf[k] = column k

    def normalize(v):                                                      # self * (1 / magnitude)
        sq = [rnd(c * c) for c in v]
        mag = sqrt_rnd(rnd(rnd(sq[0] + sq[1]) + sq[2]))
        s = rnd(Fraction(1) / mag)
        return [rnd(c * s) for c in v]

    def cross(a, b):
        return [rnd(rnd(a[1] * b[2]) - rnd(a[2] * b[1])), rnd(rnd(a[2] * b[0]) - rnd(a[0] * b[2])), rnd(rnd(a[0] * b[1]) - rnd(a[1] * b[0]))]

    def dot(a, b):
        return rnd(rnd(rnd(a[0] * b[0]) + rnd(a[1] * b[1])) + rnd(a[2] * b[2]))

    out = []
    for eye, d in zip(eyes, dirs):
        eye, f = [to_frac(c) for c in eye], normalize([to_frac(c) for c in d])
        s = normalize(cross(f, up))
        u = cross(s, f)
        view = [[s[0], u[0], -f[0], Fraction(0)], [s[1], u[1], -f[1], Fraction(0)], [s[2], u[2], -f[2], Fraction(0)],
                [-dot(eye, s), -dot(eye, u), dot(eye, f), Fraction(1)]]
        m = [[rnd(rnd(rnd(rnd(cpf[0][e] * view[k][0]) + rnd(cpf[1][e] * view[k][1])) + rnd(cpf[2][e] * view[k][2])) + rnd(cpf[3][e] * view[k][3]))
              for e in range(4)] for k in range(4)]
        out.append([[to_f32(c) for c in col] for col in m])
    return np.array(out, np.float32)


def test_cameras_oracle_equals_exact_rational_ieee_arithmetic(oracle):
    n = 40
    pos, vel = oracle.init_state(n, 11)
    rng = np.random.default_rng(11)
    pos[:, 2] = rng.uniform(-5, 5, n).astype(np.float32)
    vel[:, 2] = rng.uniform(-0.05, 0.05, n).astype(np.float32)
    up = np.array([0, 0, 1], np.float32)
    cp = oracle.camera_constant(30.0, 1.5)
    got = oracle.cameras(pos, vel, up, cp)
    ref = cameras_exact(pos, vel, up, np.asarray(cp, np.float32).reshape(4, 4))
    assert (bits(got) == bits(ref)).all()
