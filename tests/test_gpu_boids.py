"""GPU parity tests of the boids controller (update_instance_boids, src/main.rs:443-526):
the HIP kernel through the C ABI against the CPU oracle, bit for bit."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def matrices_equal(got, ref):
    """model matrices (main.rs:437-439) word for word: since round 4 the device computes the angle, its sine and cosine as the host's
    libm does (nenbody_amd/csrc/nb_libm.h); NaN entries (non-finite velocities) compare equal whatever their payload"""
    g, r = np.ascontiguousarray(got, np.float32), np.ascontiguousarray(ref, np.float32)
    return g.shape == r.shape and bool(((g.view(np.uint32) == r.view(np.uint32)) | (np.isnan(g) & np.isnan(r))).all())


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_bits_equal(got, ref, what=""):
    g, r = bits(got), bits(ref)
    assert (g == r).all(), f"{what}: {(g != r).sum()} of {g.size} words differ, first at {np.argwhere(g != r)[0]}"


def cloud(oracle, n, seed, scale=0.3):
    pos, vel = oracle.init_state(n, seed)
    rng = np.random.default_rng(seed)
    pos[:, 2] = rng.uniform(-100, 100, n).astype(np.float32)
    vel[:, 2] = rng.uniform(0, 0.1, n).astype(np.float32)
    return (pos * np.float32(scale)).astype(np.float32), vel


@pytest.fixture(params=[(0, 3), (1, 3), (2, 3), (3, 3), (4, 3), (6, 3), (0, 2), (4, 2), (2, 2), (0, 1), (1, 1), (2, 1), (3, 1), (4, 1), (6, 1),
                        (0, 4), (1, 4), (2, 4), (4, 4), (6, 4), (0, 5), (4, 5), (2, 5)],
                ids=["auto", "select", "masked3d", "select3d", "rule3-tested", "rule3-tested-3d", "packed", "packed-rule3-tested",
                     "packed-3d-tiles", "pc-auto", "pc-select", "pc-masked3d", "pc-select3d", "pc-rule3-tested", "pc-rule3-tested-3d",
                     "split", "split-select", "split-3d", "split-rule3-tested", "split-rule3-tested-3d", "split-packed",
                     "split-packed-rule3-tested", "split-packed-3d-tiles"])
def force(request, monkeypatch):
    """NB_BOIDS_FORCE: OR-ed into every tile's flags -- 1 = never the masked-FMA form, 2 = never the planar form,
    4 = always evaluate the rule-3 test (never the form that knows it holds for the whole tile);
    NB_BOIDS_PC: the launch form -- one lane per body plain (3) or with (x, y) packed (2), producer/consumer (1), two waves
    per body that split the chains, plain (4) or packed (5).
    Every form must give the same bits."""
    monkeypatch.setenv("NB_BOIDS_FORCE", str(request.param[0]))
    monkeypatch.setenv("NB_BOIDS_PC", str(request.param[1]))
    return request.param


@pytest.mark.parametrize("n", [1, 2, 3, 64, 65, 255, 256, 257, 1000, 1025, 3000])
def test_boids_ragged_sizes_bit_exact(nb, oracle, force, n):
    pos, vel = cloud(oracle, n, seed=n)
    with nb.Scene(pos, vel) as sc:
        sc.step_boids_n(3)
        p, v = sc.state()
    p_ref, v_ref = oracle.boids_run(pos, vel, 3)
    assert_bits_equal(p, p_ref, f"positions n={n}")
    assert_bits_equal(v, v_ref, f"velocities n={n}")


@pytest.mark.parametrize("tile", [256, 512, 1024])
def test_boids_every_tile_size(nb, oracle, tile):
    pos, vel = cloud(oracle, 2500, seed=tile)
    with nb.Scene(pos, vel) as sc:
        sc.step_boids_n(2, nb.default_boids_params(tile=tile))
        p, v = sc.state()
    p_ref, v_ref = oracle.boids_run(pos, vel, 2)
    assert_bits_equal(p, p_ref)
    assert_bits_equal(v, v_ref)


@pytest.mark.parametrize("n", [7, 700, 2600])
def test_boids_planar_state_every_form(nb, oracle, force, n):
    """z = 0, vz = 0 (the reference's initial state): planar shortcut on/off, masked/select forms -- same bits."""
    pos, vel = oracle.init_state(n, seed=n + 3)
    pos *= np.float32(0.2)
    with nb.Scene(pos, vel) as sc:
        sc.step_boids_n(4)
        p, v = sc.state()
    p_ref, v_ref = oracle.boids_run(pos, vel, 4)
    assert_bits_equal(p, p_ref)
    assert_bits_equal(v, v_ref)


def test_boids_mixed_tiles(nb, oracle):
    """Some tiles planar, some not, one with a non-finite record: every tile picks its own form."""
    n = 2600
    pos, vel = oracle.init_state(n, seed=99)
    pos *= np.float32(0.2)
    pos[300:500, 2] = np.linspace(-3, 3, 200, dtype=np.float32)
    vel[1500:1600, 2] = np.float32(0.05)
    pos[2000, 0] = np.inf
    with nb.Scene(pos, vel) as sc:
        sc.step_boids_n(2)
        p, v = sc.state()
    p_ref, v_ref = oracle.boids_run(pos, vel, 2)
    assert (np.isnan(p) == np.isnan(p_ref)).all() and (np.isnan(v) == np.isnan(v_ref)).all()
    ok = ~np.isnan(p_ref)
    assert (bits(p)[ok] == bits(p_ref)[ok]).all()
    okv = ~np.isnan(v_ref)
    assert (bits(v)[okv] == bits(v_ref)[okv]).all()


def test_boids_reference_initial_state_long_run(nb, oracle):
    """The reference's own planar initial distributions (main.rs:738-747), 100 steps: flocks form, speeds clamp."""
    pos, vel = nb.init_state(1024, 1234)
    with nb.Scene(pos, vel) as sc:
        sc.step_boids_n(100)
        p, v = sc.state()
        inst = sc.instances()
    p_ref, v_ref, inst_ref = oracle.boids_run(pos, vel, 100, want_instances=True)
    assert_bits_equal(p, p_ref)
    assert_bits_equal(v, v_ref)
    assert matrices_equal(inst, inst_ref)
    assert (p[:, 2] == 0).all()


def test_boids_radius_boundaries_are_exact(nb, oracle):
    """Pairs placed exactly at, just inside and just outside each radius: the squared-distance thresholds that replace
    sqrt(d2) < r on the GPU must agree with the reference's sqrt test on every one of them."""
    base = np.float32(5.0)
    xs = [np.float32(0)]
    x = base
    for _ in range(6):
        x = np.nextafter(x, np.float32(0))
    for _ in range(13):                        # 6 floats below 5.0 ... 6 above
        xs.append(x)
        x = np.nextafter(x, np.float32(10))
    r1 = np.float32(np.sqrt(1000.0))
    x = r1
    for _ in range(4):
        x = np.nextafter(x, np.float32(0))
    for _ in range(9):
        xs.append(x)
        x = np.nextafter(x, np.float32(100))
    pos = np.zeros((len(xs), 3), np.float32)
    pos[:, 0] = xs
    pos[1::2, 1] = np.float32(1e-4)            # break ties: some pairs get a tiny y offset
    vel = np.zeros_like(pos)
    vel[:, 0] = np.linspace(0, 0.05, len(xs), dtype=np.float32)
    with nb.Scene(pos, vel) as sc:
        sc.step_boids_n(1)
        p, v = sc.state()
    p_ref, v_ref = oracle.boids_run(pos, vel, 1)
    assert_bits_equal(v, v_ref)
    assert_bits_equal(p, p_ref)


def test_boids_velocity_radius_and_custom_constants(nb, oracle):
    """A rule-3 radius small enough to cut (the default 500 never does) and other non-default constants."""
    pos, vel = cloud(oracle, 800, seed=5)
    vel *= np.float32(30)                      # velocity differences up to ~4
    for r3, r2, r1 in [(1.0, 3.0, 400.0), (2.5, 0.0, 1e9), (np.inf, 7.5, 0.5), (0.5, -1.0, np.nan)]:
        bp, obp = nb.default_boids_params(), oracle.boids_params()
        for k, val in (("rule_3_distance", r3), ("rule_2_distance", r2), ("rule_1_distance", r1), ("dt", 0.1), ("rule_2_scale", 0.2)):
            setattr(bp, k, val)
            setattr(obp, k, val)
        with nb.Scene(pos, vel) as sc:
            sc.step_boids_n(2, bp)
            p, v = sc.state()
        p_ref, v_ref = oracle.boids_run(pos, vel, 2, obp)
        assert_bits_equal(v, v_ref, f"r3={r3} r2={r2} r1={r1}")
        assert_bits_equal(p, p_ref, f"r3={r3} r2={r2} r1={r1}")


@pytest.mark.parametrize("pc", ["3", "2", "4", "5"], ids=["lane-per-body", "lane-per-body-packed", "chain-split", "chain-split-packed"])
def test_boids_one_instruction_radius_tests_at_their_limits(nb, oracle, monkeypatch, pc):
    """Round 3: a radius test `d2 < T` is one instruction, clamp(fma(-d2, k, T*k)), k = 2^(30 - e_T) chosen by the host.  Radii for
    which it has no such constants (not finite; squared thresholds below 2^-90) must fall back to the compare-and-select form;
    radii at the ends of the range it does cover (tiny, huge) and data scaled to sit right at them must give the oracle's bits."""
    monkeypatch.setenv("NB_BOIDS_PC", pc)
    rng = np.random.default_rng(7)
    for scale, r1, r2, r3 in [(1e-20, 3e-40, 2e-20, 1e-19),      # r1 = 3e-40 is subnormal, (2e-20)^2 = 4e-40 too: no constants
                              (1e-12, 2e-24, 1.5e-12, 1e-11),     # squared thresholds ~ 2^-78: inside the covered range
                              (1e15, 2e30, 3e15, 5e15),           # huge: d2 ~ 1e30, k = 2^(30 - 100)
                              (3e18, 3.0e38, 1.5e19, 1.8e19),     # d2 overflows to +inf for the far pairs: the test must fail for them
                              (1.0, 1000.0, 5.0, float("inf"))]:  # an infinite radius: no constants
        n = 700
        pos = (rng.uniform(-1, 1, (n, 3)) * scale).astype(np.float32)
        vel = (rng.uniform(-1, 1, (n, 3)) * scale).astype(np.float32)
        bp, obp = nb.default_boids_params(), oracle.boids_params()
        for k, val in (("rule_1_distance", r1), ("rule_2_distance", r2), ("rule_3_distance", r3)):
            setattr(bp, k, val)
            setattr(obp, k, val)
        with nb.Scene(pos, vel) as sc:
            sc.step_boids_n(2, bp)
            p, v = sc.state()
        p_ref, v_ref = oracle.boids_run(pos, vel, 2, obp)
        nan = np.isnan(v_ref)
        assert (np.isnan(v) == nan).all() and (bits(v)[~nan] == bits(v_ref)[~nan]).all(), f"scale={scale} r1={r1} r2={r2} r3={r3}"
        nanp = np.isnan(p_ref)
        assert (np.isnan(p) == nanp).all() and (bits(p)[~nanp] == bits(p_ref)[~nanp]).all(), f"scale={scale} (positions)"


def _rule3_bound(t3):
    """the host's v_lim restated: largest binary32 V with ((2V)^2 + (2V)^2) + (2V)^2 <= t3, every operation in binary32"""
    def holds(b):
        v = np.array([b], np.uint32).view(np.float32)[0]
        with np.errstate(over="ignore"):
            w = np.float32(v + v)
            q = np.float32(w * w)
            return np.float32(np.float32(q + q) + q) <= t3
    lo, hi = 0, 0x7f7fffff
    assert holds(lo) and not holds(hi)
    while hi - lo > 1:
        mid = (lo + hi) // 2
        lo, hi = (mid, hi) if holds(mid) else (lo, mid)
    return np.array([lo], np.uint32).view(np.float32)[0]


@pytest.mark.parametrize("pc", ["3", "2", "1", "4", "5"], ids=["lane-per-body", "lane-per-body-packed", "producer-consumer", "chain-split",
                                                            "chain-split-packed"])
@pytest.mark.parametrize("planar", [True, False])
def test_boids_rule3_known_to_hold_per_tile(nb, oracle, monkeypatch, planar, pc):
    """Tiles whose velocity components all stay within the host's bound skip the rule-3 test (it cannot fail there);
    tiles with one component a single ulp above the bound evaluate it.  Velocities sit AT the bound, one ulp above it,
    and far above it (where the test does fail), so a wrong bound or a wrong flag shows as different bits."""
    monkeypatch.setenv("NB_BOIDS_PC", pc)
    n = 5000
    pos, vel = (oracle.init_state(n, 77) if planar else cloud(oracle, n, 77))
    pos *= np.float32(0.2)
    r3 = np.float32(1.0)
    t3 = np.float32(1.0)
    while np.sqrt(np.nextafter(t3, np.float32(2))) < r3:      # largest x with sqrt(x) < r3 is below 1: walk down
        t3 = np.nextafter(t3, np.float32(2))
    while not np.sqrt(t3) < r3:
        t3 = np.nextafter(t3, np.float32(0))
    vlim = _rule3_bound(t3)
    assert 0.28 < vlim < 0.29                                 # 1 / (2 sqrt 3)
    rng = np.random.default_rng(3)
    vel[:, :2] = rng.uniform(-vlim, vlim, (n, 2)).astype(np.float32)
    vel[:, 2] = 0 if planar else rng.uniform(-vlim, vlim, n).astype(np.float32)
    vel[10] = [vlim, -vlim, 0 if planar else vlim]           # at the bound: still inside
    vel[11] = [-vlim, vlim, 0 if planar else -vlim]          # the farthest pair the bound admits
    vel[1100, 0] = np.nextafter(vlim, np.float32(1))          # one ulp outside: its tile tests
    vel[2300] = [0.9, -0.8, 0]                                # fails the test against most bodies
    vel[4000:4100, 1] = np.float32(-0.6)                      # a run of bodies outside: their own workgroup tests everywhere
    bp, obp = nb.default_boids_params(), oracle.boids_params()
    bp.rule_3_distance = obp.rule_3_distance = float(r3)
    outs = {}
    for knob in ("0", "4"):
        monkeypatch.setenv("NB_BOIDS_FORCE", knob)
        with nb.Scene(pos, vel) as sc:
            sc.step_boids_n(2, bp)
            outs[knob] = sc.state()
    p_ref, v_ref = oracle.boids_run(pos, vel, 2, obp)
    for knob, (p, v) in outs.items():
        assert_bits_equal(v, v_ref, f"velocities, NB_BOIDS_FORCE={knob}")
        assert_bits_equal(p, p_ref, f"positions, NB_BOIDS_FORCE={knob}")


def test_boids_nonfinite_positions_like_the_reference(nb, oracle):
    pos, vel = cloud(oracle, 300, seed=8)
    pos[10, 0] = np.inf
    pos[20, 1] = np.nan
    with nb.Scene(pos, vel) as sc:
        sc.step_boids_n(1)
        p, v = sc.state()
    p_ref, v_ref = oracle.boids_run(pos, vel, 1)
    assert (np.isnan(p) == np.isnan(p_ref)).all() and (np.isnan(v) == np.isnan(v_ref)).all()
    ok = ~np.isnan(p_ref)
    assert (bits(p)[ok] == bits(p_ref)[ok]).all()
    okv = ~np.isnan(v_ref)
    assert (bits(v)[okv] == bits(v_ref)[okv]).all()


def test_boids_and_nbody_steps_mix(nb, oracle):
    pos, vel = cloud(oracle, 600, seed=12)
    with nb.Scene(pos, vel) as sc:
        sc.step_boids_n(2)
        sc.step_n(2)
        sc.step_boids_n(1)
        p, v = sc.state()
    pr, vr = oracle.boids_run(pos, vel, 2)
    pr, vr = oracle.run(pr, vr, 2)
    pr, vr = oracle.boids_run(pr, vr, 1)
    assert_bits_equal(p, pr)
    assert_bits_equal(v, vr)


def test_update_instance_boids_operator(nb, oracle):
    n = 400
    pos, vel = cloud(oracle, n, seed=21)
    p_ref, v_ref, inst_ref = oracle.boids_run(pos, vel, 1, want_instances=True)
    positions, velocities = pos.copy(), vel.copy()
    old_p, old_v = np.zeros_like(pos), np.zeros_like(vel)
    inst = np.zeros((n, 4, 4), np.float32)
    nb.update_instance_boids(inst, positions, old_p, velocities, old_v)
    assert_bits_equal(old_p, pos)            # main.rs:459
    assert_bits_equal(old_v, vel)            # main.rs:460
    assert_bits_equal(positions, p_ref)
    assert_bits_equal(velocities, v_ref)
    assert matrices_equal(inst, inst_ref)
    with pytest.raises(ValueError):
        nb.update_instance_boids(inst, positions, np.zeros((n - 1, 3), np.float32), velocities, old_v)


def test_update_instance_boids_other_constants_every_other_frame(nb, oracle):
    """The library keeps its device context between frames: constants that change from one frame to the next must take effect."""
    n = 500
    pos, vel = cloud(oracle, n, seed=23)
    positions, velocities = pos.copy(), vel.copy()
    old_p, old_v = np.zeros_like(pos), np.zeros_like(vel)
    inst = np.zeros((n, 4, 4), np.float32)
    p_ref, v_ref = pos, vel
    for frame in range(5):
        bp, obp = nb.default_boids_params(), oracle.boids_params()
        if frame % 2:
            bp.dt, bp.rule_2_distance, bp.rule_3_scale = 0.1, 9.0, 0.25
            obp.dt, obp.rule_2_distance, obp.rule_3_scale = 0.1, 9.0, 0.25
        nb.update_instance_boids(inst, positions, old_p, velocities, old_v, bp)
        p_ref, v_ref = oracle.boids_run(p_ref, v_ref, 1, obp)
        assert_bits_equal(positions, p_ref, f"frame {frame}")
        assert_bits_equal(velocities, v_ref, f"frame {frame}")


def test_update_instance_boids_zip_truncation_and_frames(nb, oracle):
    """instances shorter than positions: only that many bodies move while the folds see everyone (main.rs:465-471)."""
    n, m = 300, 40
    pos, vel = cloud(oracle, n, seed=22)
    p_ref, v_ref = oracle.boids_step_range(pos, vel, 0, m)
    positions, velocities = pos.copy(), vel.copy()
    inst = np.zeros((m, 4, 4), np.float32)
    nb.update_instance_boids(inst, positions, np.zeros_like(pos), velocities, np.zeros_like(vel))
    assert_bits_equal(positions[:m], p_ref)
    assert_bits_equal(velocities[:m], v_ref)
    assert_bits_equal(positions[m:], pos[m:])
    assert_bits_equal(velocities[m:], vel[m:])
    # frame after frame on the same Vecs, alternating with the n-body function (they share the cached context)
    positions, velocities = pos.copy(), vel.copy()
    old_p, old_v = np.zeros_like(pos), np.zeros_like(vel)
    inst = np.zeros((n, 4, 4), np.float32)
    p_ref, v_ref = pos, vel
    for frame in range(4):
        if frame % 2 == 0:
            nb.update_instance_boids(inst, positions, old_p, velocities, old_v)
            p_ref, v_ref = oracle.boids_run(p_ref, v_ref, 1)
        else:
            nb.update_instance_nbody(inst, positions, old_p, velocities, old_v)
            p_ref, v_ref = oracle.run(p_ref, v_ref, 1)
        assert_bits_equal(positions, p_ref, f"frame {frame}")
        assert_bits_equal(velocities, v_ref, f"frame {frame}")


@pytest.mark.parametrize("n_inst,n_pos,n_vel", [(300, 300, 40), (300, 40, 300), (25, 300, 40), (200, 180, 190), (5000, 5000, 4100),
                                                  (20000, 17000, 20000)])
def test_update_instance_boids_slices_of_unequal_length(nb, oracle, n_inst, n_pos, n_vel):
    """positions and velocities of different lengths: rule 1 and rule 2 fold over old_positions.iter() (main.rs:471, 482),
    rule 3 over old_velocities.iter() (main.rs:494) -- each its own length, neither indexed by the other's -- and the zip
    (main.rs:465-469) updates the first min(len) bodies.  Nothing panics in the reference; nothing is refused here."""
    pos, _ = cloud(oracle, n_pos, seed=41)
    _, vel = cloud(oracle, n_vel, seed=42)
    m = min(n_inst, n_pos, n_vel)
    p_ref, v_ref, i_ref = oracle.boids_update_instance(n_inst, pos, vel)
    positions, velocities = pos.copy(), vel.copy()
    old_p, old_v = np.zeros_like(pos), np.zeros_like(vel)
    inst = np.zeros((n_inst, 4, 4), np.float32)
    nb.update_instance_boids(inst, positions, old_p, velocities, old_v)
    assert_bits_equal(old_p, pos, "old_positions = copy of positions (main.rs:459)")
    assert_bits_equal(old_v, vel, "old_velocities = copy of velocities (main.rs:460)")
    assert_bits_equal(positions[:m], p_ref, "positions")
    assert_bits_equal(velocities[:m], v_ref, "velocities")
    assert_bits_equal(positions[m:], pos[m:], "bodies past the zip keep their positions")
    assert_bits_equal(velocities[m:], vel[m:], "bodies past the zip keep their velocities")
    assert np.abs(inst[:m, 3, :3] - p_ref).max() == 0.0
    assert matrices_equal(inst[:m], i_ref)
    assert not inst[m:].any()


def test_boids_sharded_launch_equals_unsharded(nb, oracle):
    """What each rank of a multi-GPU job runs: index ranges of one step, positions AND velocities written per range."""
    import torch

    from nenbody_amd.dist import HipBackend

    be = HipBackend()
    n = 1500
    pos, vel = cloud(oracle, n, seed=31)
    dev = torch.device("cuda", 0)

    def rec(a):
        t = torch.zeros((n, 4), dtype=torch.float32)
        t[:, :3] = torch.from_numpy(a)
        return t.to(dev)

    pin, vin = rec(pos), rec(vel)
    pout, vout = torch.zeros_like(pin), torch.zeros_like(vin)
    bp = nb.default_boids_params()
    for first, count in [(0, 1), (1, 255), (256, 700), (956, 544)]:
        be.boids_step(bp, n, first, count, pin, vin, pout, vout)
    torch.cuda.synchronize()
    p_ref, v_ref = oracle.boids_run(pos, vel, 1)
    assert_bits_equal(pout[:, :3].cpu().numpy(), p_ref)
    assert_bits_equal(vout[:, :3].cpu().numpy(), v_ref)


def test_boids_full_size_sampled_vs_oracle(nb, oracle):
    """N = 131 072 (BASELINE size): sampled bodies folded over all j by the oracle."""
    n = 131072
    pos, vel = nb.init_state(n, 1234)
    with nb.Scene(pos, vel) as sc:
        sc.step_boids_n(1)
        p, v = sc.state()
    idx = np.unique(np.concatenate([np.arange(0, 16), np.arange(n - 16, n), np.linspace(16, n - 17, 64).astype(np.int64)]))
    for i in idx:
        p_ref, v_ref = oracle.boids_step_range(pos, vel, int(i), 1)
        assert (bits(p[i]) == bits(p_ref[0])).all() and (bits(v[i]) == bits(v_ref[0])).all(), f"body {i}"
    speed = np.sqrt((v.astype(np.float64) ** 2).sum(axis=1))
    assert speed.max() <= 1.0 + 1e-6            # the clamp of main.rs:516-518


# ---------------------------------------------------------------------------------------------------------------------------
# the split form (round 4, nb_launch_boids_step_split): the j range in slices, one lane per body per slice, the slices' sums added
# in slice order -- the reference's predicates on the reference's operands (same neighbour sets, same counts), reassociated sums
# ---------------------------------------------------------------------------------------------------------------------------
def split_step(nb, pos, vel, parts, bp=None):
    """one boids step of the set, every (first, count) of `parts` through the split form on the one GPU"""
    import torch

    from nenbody_amd.dist import HipBackend

    be, dev, n = HipBackend(), torch.device("cuda", 0), len(pos)
    bp = bp if bp is not None else nb.default_boids_params()

    def rec(a):
        t = torch.zeros((n, 4), dtype=torch.float32)
        t[:, :3] = torch.from_numpy(a)
        return t.to(dev)

    pin, vin = rec(pos), rec(vel)
    pout, vout = torch.full_like(pin, float("nan")), torch.full_like(vin, float("nan"))
    for first, count in parts:
        scratch = torch.empty((max(16, be.boids_split_scratch_bytes(bp, n, count)),), dtype=torch.uint8, device=dev)
        be.boids_step_split(bp, n, first, count, pin, vin, pout, vout, scratch)
    torch.cuda.synchronize()
    return pout[:, :3].cpu().numpy(), vout[:, :3].cpu().numpy()


def boids_velocity_f64(pos, vel, i, bp):
    """the new velocity of body i (main.rs:471-518) with the reference's binary32 PREDICATES -- the same neighbour sets -- and every
    sum, mean and blend carried in binary64: the yardstick for the rounding error of a binary32 sum, the reference's included"""
    f = np.float32
    d = pos - pos[i]
    d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    other = np.arange(len(pos)) != i
    with np.errstate(invalid="ignore"):
        p1 = (d2 < f(bp.rule_1_distance)) & other
        p2 = (np.sqrt(d2) < f(bp.rule_2_distance)) & other
        e = vel - vel[i]
        e2 = (e[:, 0] * e[:, 0] + e[:, 1] * e[:, 1]) + e[:, 2] * e[:, 2]
        p3 = (np.sqrt(e2) < f(bp.rule_3_distance)) & other
    c = pos[p1].astype(np.float64).sum(axis=0)
    r = -(d[p2].astype(np.float64)).sum(axis=0)
    m = vel[p3].astype(np.float64).sum(axis=0)
    if p1.sum():
        c = c / p1.sum()
    if p3.sum():
        m = m / p3.sum()
    v = c * float(f(bp.rule_1_scale)) + r * float(f(bp.rule_2_scale)) + m * float(f(bp.rule_3_scale))
    mag = np.sqrt((v * v).sum())
    return v / mag if mag > 1.0 else v


def close_to_the_reference(v, v_ref, p, p_ref, what="", state=None):
    """The split form against the bit-exact step.  Its neighbour sets and counts are the reference's; its sums are binary32 sums in
    another order, and a sequential binary32 sum of m terms carries ~sqrt(m) half-ulps of its own (6.9e-6 of a unit velocity on
    20 000 bodies in a 40 x 40 square), so "close" means: every body within 5e-5 of the largest velocity component, and -- with
    `state` = (pos, vel, bp) -- on the 48 bodies where the two differ most, plus 16 evenly spaced ones, the split form is NO FURTHER
    from the same sums carried in binary64 than the reference's own arithmetic is: worst body against worst body (plus four
    ulps of the result) and on average (body by body the two errors are independent: either may be the larger)."""
    sv = float(np.abs(v_ref).max())
    dv = np.abs(v - v_ref).max(axis=1)
    assert dv.max() <= 5e-5 * sv, f"{what}: max |dv| {dv.max():.3e} against {5e-5 * sv:.3e}"
    assert np.abs(p - p_ref).max() <= 5e-5 * sv + float(np.spacing(np.float32(np.abs(p_ref).max()))), what
    if state is None:
        return
    pos, vel, bp = state
    chosen = np.unique(np.concatenate([np.argsort(dv)[-48:], np.linspace(0, len(v) - 1, 16).astype(np.int64)]))
    ulp = float(np.spacing(np.float32(sv)))
    v64 = np.array([boids_velocity_f64(pos, vel, int(i), bp) for i in chosen])
    err_split, err_ref = np.abs(v[chosen] - v64).max(axis=1), np.abs(v_ref[chosen] - v64).max(axis=1)
    # the worst body against the reference's worst, and body by body with the slack of two independent roundings of a mean
    assert err_split.max() <= err_ref.max() + 4 * ulp, f"{what}: split {err_split.max():.3e} from the binary64 sums, the reference {err_ref.max():.3e}"
    assert err_split.mean() <= 1.5 * err_ref.mean() + ulp, f"{what}: mean error split {err_split.mean():.3e}, the reference {err_ref.mean():.3e}"


@pytest.mark.parametrize("slices", ["1", "2", "3", "7", "auto"])
@pytest.mark.parametrize("n,parts", [(64, [(0, 64)]), (300, [(0, 300)]), (1025, [(0, 1), (1, 1024)]), (3000, [(0, 1000), (1000, 2000)]),
                                     (20000, [(0, 2500), (2500, 17500)])], ids=lambda x: str(x) if isinstance(x, int) else "")
def test_boids_split_form_vs_oracle(nb, oracle, monkeypatch, n, parts, slices):
    if slices != "auto":
        monkeypatch.setenv("NB_BOIDS_SLICES", slices)
    for three_d in (True, False):
        pos, vel = cloud(oracle, n, seed=n + 1) if three_d else (lambda pv: (pv[0] * np.float32(0.2), pv[1]))(oracle.init_state(n, n + 2))
        p, v = split_step(nb, pos, vel, parts)
        p_ref, v_ref = oracle.boids_run(pos, vel, 1)
        if slices == "1":            # one slice with rule 3 kept in the loop (NB_BOIDS_FORCE=4: otherwise its sum is the total of the
            monkeypatch.setenv("NB_BOIDS_FORCE", "4")   # velocities minus the body's own): the reference's order of additions, hence its bits
            pb, vb = split_step(nb, pos, vel, parts)
            monkeypatch.delenv("NB_BOIDS_FORCE")
            assert_bits_equal(vb, v_ref, "one slice")
            assert_bits_equal(pb, p_ref, "one slice")
        close_to_the_reference(v, v_ref, p, p_ref, f"n={n} slices={slices} 3d={three_d}", (pos, vel, nb.default_boids_params()))
        p2, v2 = split_step(nb, pos, vel, parts)
        assert_bits_equal(v, v2, "run to run")


@pytest.mark.parametrize("knob", ["1", "2", "4", "6", "7"], ids=["select", "3d", "rule3-tested", "rule3-tested-3d", "select-3d-tested"])
@pytest.mark.parametrize("tile", [256, 512, 1024])
def test_boids_split_form_every_tile_form(nb, oracle, monkeypatch, knob, tile):
    monkeypatch.setenv("NB_BOIDS_FORCE", knob)
    monkeypatch.setenv("NB_BOIDS_SLICES", "5")
    n = 6000
    pos, vel = cloud(oracle, n, seed=77)
    pos[300:500, 2] = 0
    bp = nb.default_boids_params(tile=tile)
    p, v = split_step(nb, pos, vel, [(0, 2000), (2000, 4000)], bp)
    p_ref, v_ref = oracle.boids_run(pos, vel, 1)
    close_to_the_reference(v, v_ref, p, p_ref, f"force={knob} tile={tile}", (pos, vel, bp))


def test_boids_split_form_radius_boundaries_and_custom_constants(nb, oracle, monkeypatch):
    """the radius tests decide WHO is in a sum: at, just inside and just outside every radius the split form must make the
    reference's choice (a wrong neighbour moves a mean by far more than a rounding), also with radii that cut"""
    monkeypatch.setenv("NB_BOIDS_SLICES", "4")
    xs, x = [np.float32(0)], np.float32(5.0)
    for _ in range(6):
        x = np.nextafter(x, np.float32(0))
    for _ in range(13):
        xs.append(x)
        x = np.nextafter(x, np.float32(10))
    x = np.float32(np.sqrt(1000.0))
    for _ in range(4):
        x = np.nextafter(x, np.float32(0))
    for _ in range(9):
        xs.append(x)
        x = np.nextafter(x, np.float32(100))
    reps = 60                                    # the boundary pairs spread over several 256-record tiles and slices
    pos = np.zeros((len(xs) * reps, 3), np.float32)
    pos[:, 0] = np.tile(xs, reps)
    pos[:, 1] = np.repeat(np.arange(reps, dtype=np.float32) * np.float32(200.0), len(xs))   # replicas 200 apart: only their own pairs interact
    pos[1::2, 1] += np.float32(1e-4)
    vel = np.zeros_like(pos)
    vel[:, 0] = np.linspace(0, 0.05, len(pos), dtype=np.float32)
    bp = nb.default_boids_params(tile=256)
    p, v = split_step(nb, pos, vel, [(0, len(pos))], bp)
    p_ref, v_ref = oracle.boids_run(pos, vel, 1)
    close_to_the_reference(v, v_ref, p, p_ref, "boundaries", (pos, vel, bp))
    pos, vel = cloud(oracle, 800, seed=5)
    vel *= np.float32(30)
    for r3, r2, r1 in [(1.0, 3.0, 400.0), (2.5, 0.0, 1e9), (np.inf, 7.5, 0.5), (0.5, -1.0, np.nan)]:
        bp, obp = nb.default_boids_params(tile=256), oracle.boids_params()
        for k, val in (("rule_3_distance", r3), ("rule_2_distance", r2), ("rule_1_distance", r1), ("dt", 0.1), ("rule_2_scale", 0.2)):
            setattr(bp, k, val)
            setattr(obp, k, val)
        p, v = split_step(nb, pos, vel, [(0, 800)], bp)
        p_ref, v_ref = oracle.boids_run(pos, vel, 1, obp)
        close_to_the_reference(v, v_ref, p, p_ref, f"r3={r3} r2={r2} r1={r1}", (pos, vel, bp))


def test_boids_split_form_nonfinite_records(nb, oracle, monkeypatch):
    monkeypatch.setenv("NB_BOIDS_SLICES", "3")
    n = 2600
    pos, vel = oracle.init_state(n, seed=99)
    pos *= np.float32(0.2)
    pos[2000, 0] = np.inf
    vel[100, 1] = np.nan
    p, v = split_step(nb, pos, vel, [(0, n)])
    p_ref, v_ref = oracle.boids_run(pos, vel, 1)
    assert (np.isnan(p) == np.isnan(p_ref)).all() and (np.isnan(v) == np.isnan(v_ref)).all()
    assert (np.isinf(p) == np.isinf(p_ref)).all() and (p[np.isinf(p_ref)] == p_ref[np.isinf(p_ref)]).all()
    ok = np.isfinite(v_ref).all(axis=1) & np.isfinite(p_ref).all(axis=1)
    close_to_the_reference(v[ok], v_ref[ok], p[ok], p_ref[ok], "finite bodies")


def test_boids_split_form_at_the_headline_size(nb, oracle):
    """every rank's (first, count) of an 8-rank job at N = 131 072 through the split form (the library's own slices), sampled bodies
    of every rank against the oracle; then ShardedScene.step_boids(split=True) on a world of one"""
    n = 131072
    pos, vel = nb.init_state(n, 1234)
    parts = nb.partition(n, 8)
    p, v = split_step(nb, pos, vel, parts)
    idx = np.unique(np.concatenate([[f, f + c - 1, f + c // 2, f + c // 3] for f, c in parts]))
    sv = 0.0
    refs = [oracle.boids_step_range(pos, vel, int(i), 1) for i in idx]
    sv = max(float(np.abs(r[1]).max()) for r in refs)
    bp = nb.default_boids_params()
    ulp = float(np.spacing(np.float32(sv)))
    for i, (p_ref, v_ref) in zip(idx, refs):
        assert np.abs(v[i] - v_ref[0]).max() <= 5e-5 * sv and np.abs(p[i] - p_ref[0]).max() <= 1e-5, f"body {i}"
        v64 = boids_velocity_f64(pos, vel, int(i), bp)
        assert np.abs(v[i] - v64).max() <= np.abs(v_ref[0] - v64).max() + 2e-6 * sv + 0 * ulp, f"body {i}: further from the binary64 sums than the reference"
    sc = nb.ShardedScene(pos[:4096], vel[:4096])
    sc.step_boids(split=True)
    sc.sync()
    p_ref, v_ref = oracle.boids_run(pos[:4096], vel[:4096], 1)
    close_to_the_reference(sc.velocities(), v_ref, sc.positions(), p_ref, "ShardedScene(split=True)")
