"""Randomised differential tests on the GPU: many small random problems (sizes, shard ranges, constants, data scales,
launch shapes) -- STRICT n-body and boids against the oracle bit for bit, FAST against a tolerance.  Seeded, so a failure
is reproducible from the printed case."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# NB_RANDOM_CASES=N widens every family below to N seeded cases (a soak: `NB_RANDOM_CASES=1500 pytest tests/test_gpu_random_differential.py`)
CASES = int(os.environ.get("NB_RANDOM_CASES", "36"))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def random_state(rng, n):
    scale = float(rng.choice([1e-3, 0.1, 1.0, 30.0, 1e3, 1e5]))
    pos = (rng.uniform(-1, 1, (n, 3)) * scale).astype(np.float32)
    vel = (rng.uniform(-1, 1, (n, 3)) * float(rng.choice([0.0, 1e-3, 0.1, 2.0]))).astype(np.float32)
    kind = rng.integers(0, 4)
    if kind == 0:        # planar, like the reference's initial state
        pos[:, 2] = 0
        vel[:, 2] = 0
    elif kind == 1:      # clustered: many coincident and nearly coincident bodies
        pos[rng.integers(0, n, n // 2)] = pos[rng.integers(0, n, n // 2)]
    elif kind == 2 and n > 4:   # a few outliers far outside the STRICT ladder's range
        pos[rng.integers(0, n, 3)] *= np.float32(1e9)
        pos[rng.integers(0, n, 2)] *= np.float32(1e-30)
    return pos, vel


@pytest.mark.parametrize("case", range(CASES))
def test_strict_random_problems_bit_exact(nb, oracle, monkeypatch, case):
    rng = np.random.default_rng(1000 + case)
    n = int(rng.choice([1, 2, 5, 63, 64, 65, 200, 257, 700, 1500, 2300]))
    k = int(rng.integers(1, 4))
    shape = rng.choice(["auto", "lanes1", "lanes2", "lanes4", "lanes8", "lanes16", "pc8", "pc14", "bc", "sl", "sl"])
    if shape == "bc":
        monkeypatch.setenv("NB_STRICT_BC", "1")
    elif shape == "sl":       # the scalar-load form whole sets take (nb_nbody_sl.inc), one of its three launch shapes
        monkeypatch.setenv("NB_STRICT_SL", str(int(rng.integers(1, 4))))
        monkeypatch.setenv("NB_STRICT_BC", "0")
    elif shape.startswith("pc"):
        monkeypatch.setenv("NB_STRICT_PC", shape[2:])
    elif shape.startswith("lanes"):
        monkeypatch.setenv("NB_STRICT_PC", "0")
        monkeypatch.setenv("NB_STRICT_LANES", shape[5:])
    p = nb.default_params()
    p.dt = float(rng.choice([0.1, 0.01, 1.0]))
    p.G = float(rng.choice([0.001, 1.0, -0.05, 1e-6]))
    p.bias = float(rng.choice([1e-7, 1e-3, 2.0]))
    p.tile = int(rng.choice([0, 256, 1024]))
    pos, vel = random_state(rng, n)
    with nb.Scene(pos, vel, p) as sc:
        sc.step_n(k)
        got_p, got_v = sc.state()
    ref_p, ref_v = oracle.run(pos, vel, k, np.float32(p.dt), np.float32(p.G), np.float32(p.bias))
    what = f"case {case}: n={n} k={k} shape={shape} dt={p.dt} G={p.G} bias={p.bias} tile={p.tile}"
    nan = np.isnan(ref_p)
    assert (np.isnan(got_p) == nan).all(), what
    assert (bits(got_p)[~nan] == bits(ref_p)[~nan]).all(), what
    nanv = np.isnan(ref_v)
    assert (bits(got_v)[~nanv] == bits(ref_v)[~nanv]).all(), what


@pytest.mark.parametrize("case", range(CASES))
def test_boids_random_problems_bit_exact(nb, oracle, monkeypatch, case):
    rng = np.random.default_rng(2000 + case)
    n = int(rng.choice([1, 3, 64, 100, 256, 300, 900, 1300, 2100]))
    k = int(rng.integers(1, 4))
    monkeypatch.setenv("NB_BOIDS_FORCE", str(int(rng.integers(0, 8))))   # select form / never planar / rule 3 always tested
    monkeypatch.setenv("NB_BOIDS_PC", str(int(rng.integers(0, 6))))      # every launch form (nb_api.hip:boids_form)
    bp, obp = nb.default_boids_params(tile=int(rng.choice([0, 256, 512]))), oracle.boids_params()
    for name, choices in (("rule_1_distance", [1000.0, 50.0, 1e6]), ("rule_2_distance", [5.0, 0.5, 40.0]),
                          ("rule_3_distance", [500.0, 0.3, 2.0]), ("dt", [0.04, 0.5]), ("rule_1_scale", [0.02, -0.01]),
                          ("rule_2_scale", [0.05, 1.0]), ("rule_3_scale", [0.5, 0.0])):
        val = float(rng.choice(choices))
        setattr(bp, name, val)
        setattr(obp, name, val)
    pos, vel = random_state(rng, n)
    with nb.Scene(pos, vel) as sc:
        sc.step_boids_n(k, bp)
        got_p, got_v = sc.state()
    ref_p, ref_v = oracle.boids_run(pos, vel, k, obp)
    what = f"case {case}: n={n} k={k} NB_BOIDS_FORCE={os.environ['NB_BOIDS_FORCE']} NB_BOIDS_PC={os.environ['NB_BOIDS_PC']}"
    nan = np.isnan(ref_p)
    assert (np.isnan(got_p) == nan).all(), what
    assert (bits(got_p)[~nan] == bits(ref_p)[~nan]).all(), what
    nanv = np.isnan(ref_v)
    assert (bits(got_v)[~nanv] == bits(ref_v)[~nanv]).all(), what


@pytest.mark.parametrize("case", range(12))
def test_fast_random_problems_within_tolerance(nb, oracle, monkeypatch, case):
    rng = np.random.default_rng(3000 + case)
    n = int(rng.choice([64, 300, 1000, 2500, 4000]))
    if case >= 8:                                                                # the pairs form: whole 256-body blocks
        n = 256 * int(rng.integers(1, 20))
        monkeypatch.setenv("NB_FAST_PAIRS", "1")
        monkeypatch.setenv("NB_FAST_PAIRS_W", str(int(rng.choice([1, 2, 4, 8]))))
    else:
        monkeypatch.setenv("NB_FAST_IB", str(int(rng.choice([1, 2, 4]))))
        monkeypatch.setenv("NB_FAST_SLICES", str(int(rng.choice([1, 3, 8]))))
        if case % 4 < 2:
            monkeypatch.setenv("NB_FAST_GROUPS", str(int(rng.choice([1, 2, 4]))))    # an LDS form
        else:
            monkeypatch.setenv("NB_FAST_SL", "1")                                    # the scalar-load form
    pos = (rng.uniform(-100, 100, (n, 3))).astype(np.float32)
    vel = (rng.uniform(0, 0.1, (n, 3))).astype(np.float32)
    if case % 2 == 0:
        pos[:, 2] = 0
        vel[:, 2] = 0
    with nb.Scene(pos, vel, nb.default_params(mode=nb.NB_MODE_FAST)) as sc:
        sc.step_n(1)
        got_p, got_v = sc.state()
    ref_p, ref_v = oracle.run(pos, vel, 1)
    acc = np.abs(ref_v - vel).max()
    # relative force error per step, plus one ulp of the velocity itself (v + a*dt is rounded after the sum)
    assert np.abs(got_v - ref_v).max() <= 2e-5 * acc + 1.2e-7 * np.abs(ref_v).max() + 1e-9, f"case {case}: n={n}"
    assert np.abs(got_p - ref_p).max() <= 1e-5, f"case {case}"


@pytest.mark.parametrize("case", range(max(8, CASES // 3)))
def test_fast_pairs_form_random_problems(nb, oracle, monkeypatch, case):
    """the FAST pairs form over random block counts, workgroup widths and chunkings (one tile, several, a ragged last chunk),
    planar and 3-D: within FAST's tolerance of the oracle and bit-identical from run to run"""
    rng = np.random.default_rng(7000 + case)
    np_ = int(rng.choice([2, 4]))                                   # packed pairs of bodies per lane: blocks of 256 or 512
    w = int(rng.choice([1, 2, 4, 8] if np_ == 2 else [1, 2, 4]))
    n = 128 * np_ * int(rng.integers(1, 56 // np_))
    monkeypatch.setenv("NB_FAST_PAIRS", "1")
    monkeypatch.setenv("NB_FAST_PAIRS_NP", str(np_))
    monkeypatch.setenv("NB_FAST_PAIRS_W", str(w))
    if case % 3:
        monkeypatch.setenv("NB_FAST_PAIRS_CHUNK", str(128 * np_ * w * int(rng.integers(1, 5))))
    pos = (rng.uniform(-100, 100, (n, 3))).astype(np.float32)
    vel = (rng.uniform(0, 0.1, (n, 3))).astype(np.float32)
    if case % 2 == 0:
        pos[:, 2] = 0
        vel[:, 2] = 0
    outs = []
    for _ in range(2):
        with nb.Scene(pos, vel, nb.default_params(mode=nb.NB_MODE_FAST)) as sc:
            sc.step_n(1)
            outs.append(sc.state())
    assert (bits(outs[0][0]) == bits(outs[1][0])).all() and (bits(outs[0][1]) == bits(outs[1][1])).all(), f"case {case}: not deterministic"
    got_p, got_v = outs[0]
    ref_p, ref_v = oracle.run(pos, vel, 1)
    acc = np.abs(ref_v - vel).max()
    assert np.abs(got_v - ref_v).max() <= 2e-5 * acc + 1.2e-7 * np.abs(ref_v).max() + 1e-9, f"case {case}: n={n} w={w} np={np_}"
    assert np.abs(got_p - ref_p).max() <= 1e-5, f"case {case}"


@pytest.mark.parametrize("case", range(max(12, CASES // 2)))
def test_fast_pairs_form_on_shards_random_problems(nb, oracle, monkeypatch, case):
    """the FAST pairs form across the ranks of a multi-GPU job (nb_launch_ring_fold / _finish; every rank's launches on the one GPU,
    the second exchange by hand) over random rank counts, blocks per rank, bodies per lane, a-blocks per launch and waves per a-block,
    planar / 3-D / partly planar data, other constants: within FAST's tolerance of the oracle, identical bits from run to run, and
    -- the same pairs, evaluated once on one side instead of twice -- close to the one-GPU FAST step"""
    from test_gpu_ring import ring_steps_on_one_gpu

    rng = np.random.default_rng(9000 + case)
    np_ = int(rng.choice([2, 4]))
    world = int(rng.choice([2, 3, 4, 5, 6, 7, 8, 16]))
    nb_per_rank = int(rng.integers(1, 7 if np_ == 4 else 10))
    n = 128 * np_ * nb_per_rank * world
    monkeypatch.setenv("NB_RING", "1")
    monkeypatch.setenv("NB_RING_NP", str(np_))
    if case % 3 == 1:
        monkeypatch.setenv("NB_RING_GA", str(int(rng.integers(1, nb_per_rank + 1))))
    if case % 4 >= 2:
        monkeypatch.setenv("NB_RING_WPB", str(4 * int(rng.integers(1, 9))))
    if case % 5 == 4:
        monkeypatch.setenv("NB_FAST_NO_SHARE", "1")
    p = nb.default_params(mode=nb.NB_MODE_FAST)
    p.dt = float(rng.choice([0.1, 0.01]))
    p.G = float(rng.choice([0.001, 1.0, -0.05]))
    p.bias = float(rng.choice([1e-7, 1e-3, 2.0]))
    pos = (rng.uniform(-100, 100, (n, 3))).astype(np.float32)
    vel = (rng.uniform(0, 0.1, (n, 3))).astype(np.float32)
    if case % 2 == 0:
        pos[:, 2] = 0
        vel[:, 2] = 0
    elif case % 4 == 1:
        pos[: n // 2, 2] = 0         # planar a sides meet 3-D b sides: one flag for the step
    what = f"case {case}: n={n} world={world} np={np_} blocks/rank={nb_per_rank} G={p.G} bias={p.bias}"
    got_p, got_v = ring_steps_on_one_gpu(nb, pos, vel, world, p, 1)
    again_p, again_v = ring_steps_on_one_gpu(nb, pos, vel, world, p, 1)
    assert (bits(got_p) == bits(again_p)).all() and (bits(got_v) == bits(again_v)).all(), what + ": not deterministic"
    ref_p, ref_v = oracle.run(pos, vel, 1, np.float32(p.dt), np.float32(p.G), np.float32(p.bias))
    acc = np.abs(ref_v - vel).max()
    assert np.abs(got_v - ref_v).max() <= 2e-5 * acc + 1.2e-7 * np.abs(ref_v).max() + 1e-9, what
    assert np.abs(got_p - ref_p).max() <= 2e-5 * acc + 1.6e-5, what


@pytest.mark.parametrize("case", range(max(12, CASES // 2)))
def test_fast_pairs_form_in_phases_random_problems(nb, oracle, monkeypatch, case):
    """the same form with its step in PHASES (nb_launch_ring_fold_phase: pairs inside a rank's own slot apart from all others, the
    first phase given a snapshot whose other slots are NaN) over random rank counts, blocks per rank, bodies per lane, sub-tiles per
    workgroup of both phases and caps of the first (none of it, a part, all of a block's own part), planar / 3-D / partly planar data
    -- planar own slots inside a 3-D set and the other way round --, other constants: within FAST's tolerance of the oracle and of
    the one-launch form, identical bits from run to run"""
    from test_gpu_ring import ring_steps_on_one_gpu

    rng = np.random.default_rng(12000 + case)
    np_ = int(rng.choice([2, 4]))
    world = int(rng.choice([2, 3, 4, 5, 6, 7, 8, 16]))
    nb_per_rank = int(rng.integers(1, 7 if np_ == 4 else 10))
    n = 128 * np_ * nb_per_rank * world
    monkeypatch.setenv("NB_RING", "1")
    monkeypatch.setenv("NB_RING_NP", str(np_))
    if case % 3 != 0:
        monkeypatch.setenv("NB_RING_C4_OWN", str(int(rng.integers(1, 12))))
    if case % 3 == 1:
        monkeypatch.setenv("NB_RING_C4_REST", str(int(rng.integers(1, 40))))
    if case % 4 != 3:
        monkeypatch.setenv("NB_RING_CAP", str(int(rng.choice([1, 2, 4, 7, 8, 16, 24, 100000]))))
    if case % 5 == 4:
        monkeypatch.setenv("NB_FAST_NO_SHARE", "1")
    p = nb.default_params(mode=nb.NB_MODE_FAST)
    p.dt = float(rng.choice([0.1, 0.01]))
    p.G = float(rng.choice([0.001, 1.0, -0.05]))
    p.bias = float(rng.choice([1e-7, 1e-3, 2.0]))
    pos = (rng.uniform(-100, 100, (n, 3))).astype(np.float32)
    vel = (rng.uniform(0, 0.1, (n, 3))).astype(np.float32)
    S = n // world
    if case % 2 == 0:
        pos[:, 2] = 0
        vel[:, 2] = 0
        if case % 4 == 2:            # one rank's slot 3-D inside a planar set: its own-slot phase alone takes the 3-D path
            r = int(rng.integers(0, world))
            pos[r * S:(r + 1) * S, 2] = rng.uniform(-100, 100, S).astype(np.float32)
    elif case % 4 == 1:
        pos[: n // 2, 2] = 0         # planar own slots inside a 3-D set: the second phase sweeps them on the 3-D path
    what = f"case {case}: n={n} world={world} np={np_} blocks/rank={nb_per_rank} G={p.G} bias={p.bias}"
    got_p, got_v = ring_steps_on_one_gpu(nb, pos, vel, world, p, 1, phases=True)
    again_p, again_v = ring_steps_on_one_gpu(nb, pos, vel, world, p, 1, phases=True)
    assert (bits(got_p) == bits(again_p)).all() and (bits(got_v) == bits(again_v)).all(), what + ": not deterministic"
    if case % 3 == 0:   # the fused finish, two steps (the second starts on the planes and verdicts the first's finish left): the same bits
        two = ring_steps_on_one_gpu(nb, pos, vel, world, p, 2, phases=True)
        two_fused = ring_steps_on_one_gpu(nb, pos, vel, world, p, 2, phases="fused")
        assert all((bits(x) == bits(y)).all() for x, y in zip(two, two_fused)), what + ": the fused finish"
    ref_p, ref_v = oracle.run(pos, vel, 1, np.float32(p.dt), np.float32(p.G), np.float32(p.bias))
    acc = np.abs(ref_v - vel).max()
    assert np.isfinite(got_v).all(), what + ": a slot in flight was read"
    assert np.abs(got_v - ref_v).max() <= 2e-5 * acc + 1.2e-7 * np.abs(ref_v).max() + 1e-9, what
    assert np.abs(got_p - ref_p).max() <= 2e-5 * acc + 1.6e-5, what
    one_p, one_v = ring_steps_on_one_gpu(nb, pos, vel, world, p, 1)
    assert np.abs(got_v - one_v).max() <= 2e-5 * acc + 1.2e-7 * np.abs(ref_v).max() + 1e-9, what + ": against the one-launch form"
