"""GPU parity tests of the pairs form on SHARDS ("half shell": nb_launch_ring_fold / nb_launch_ring_finish, include/nenbody.h):
every rank of a 2-, 3-, 4- or 8-rank job evaluates the pairs of its bodies with the half of the ring that follows them, the
other bodies' halves travel in a second exchange.  Here every rank's launches run one after another on the one GPU, the
exchange is done by hand (chunk d of rank r's sums -> chunk d - 1 of rank r + d's receive buffer), and the assembled step is
held to the CPU oracle (the reference's arithmetic, main.rs:425-436) at FAST's tolerances and to the same sum in binary64.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def ring_steps_on_one_gpu(nb, pos, vel, world, params, steps=1, keep=None, phases=False):
    """`steps` steps of an n-body set cut into `world` equal ranks, every rank's fold and finish on the one GPU, the second
    exchange and the all-gather by hand.  Returns (positions, velocities).  keep: a dict that receives the ranks' `sums`.
    phases: False = nb_launch_ring_fold; True = the step in phases (nb_launch_ring_fold_phase) in the order a host with the
    exchanges hidden issues them -- and the phase that may run while the all-gather is still landing gets a snapshot whose OTHER
    slots are NaN: reading a record that could still be in flight poisons the step.  "fused": the phases with the fused finish
    (nb_launch_ring_fold_phase(NB_RING_OWN_READY) / nb_launch_ring_finish_phase): from the second step on the first phase gets NO
    positions at all (all NaN: its planes are what the finish before left), odd steps add the rank's own records in the finish
    (records [0, S) of `sums` stay poisoned), even steps run NB_RING_SUMS and hand its records over."""
    import torch

    from nenbody_amd.dist import HipBackend

    be, dev = HipBackend(), torch.device("cuda", 0)
    n = len(pos)
    S = n // world
    assert S * world == n
    cur = torch.zeros((n, 4), dtype=torch.float32)
    cur[:, :3] = torch.from_numpy(pos)
    cur = cur.to(dev)
    nxt = torch.zeros_like(cur)
    D = be.ring_partners(params, n, 0, S)
    assert D >= 1 and all(be.ring_partners(params, n, r * S, S) == D for r in range(world))
    vels, sums, recv, scratch = [], [], [], []
    for r in range(world):
        vr = torch.zeros((S, 4), dtype=torch.float32)
        vr[:, :3] = torch.from_numpy(vel[r * S:(r + 1) * S])
        vels.append(vr.to(dev))
        sums.append(torch.full(((D + 1) * S, 4), float("nan"), device=dev))   # every record must be written by the fold
        recv.append(torch.full((D * S, 4), float("nan"), device=dev))
        scratch.append(torch.empty((be.ring_scratch_bytes(params, n, r * S, S),), dtype=torch.uint8, device=dev))
    L = nb._lib
    fused = phases == "fused"
    for step in range(steps):
        in_finish = fused and step % 2 == 0   # the finish adds the rank's own records itself
        for r in range(world):
            if not phases:
                be.ring_fold(params, n, r * S, S, cur, sums[r], scratch[r])
                continue
            assert be.ring_phased(params, n, r * S, S)
            only_mine = torch.full_like(cur, float("nan"))
            a = (params, n, r * S, S)
            if fused and step:
                be.ring_fold_phase(*a, L.NB_RING_OWN_READY, only_mine, sums[r], scratch[r])
            else:
                only_mine[r * S:(r + 1) * S] = cur[r * S:(r + 1) * S]
                be.ring_fold_phase(*a, L.NB_RING_OWN, only_mine, sums[r], scratch[r])
            be.ring_fold_phase(*a, L.NB_RING_REST, cur, sums[r], scratch[r])
            sent = sums[r][S:].clone()   # the sums of the ranks in front are final behind NB_RING_REST: what a host would send now
            if in_finish:
                sums[r][:S] = float("nan")
            else:
                be.ring_fold_phase(*a, L.NB_RING_SUMS, cur, sums[r], scratch[r])
            assert (sent.view(torch.int32) == sums[r][S:].view(torch.int32)).all(), "a later phase touched what had been sent"
        for r in range(world):
            for d in range(1, D + 1):
                recv[r][(d - 1) * S:d * S] = sums[(r - d) % world][d * S:(d + 1) * S]
        for r in range(world):
            if fused:
                be.ring_finish_phase(params, n, r * S, S, cur, nxt, vels[r], None if in_finish else sums[r], recv[r], scratch[r])
            else:
                be.ring_finish(params, n, r * S, S, cur, nxt, vels[r], sums[r], recv[r])
        cur, nxt = nxt, cur
    torch.cuda.synchronize()
    if keep is not None:
        keep["sums"] = [s.cpu().numpy() for s in sums]
    return cur[:, :3].cpu().numpy(), np.concatenate([v[:, :3].cpu().numpy() for v in vels])


def state(oracle, n, seed, three_d):
    pos, vel = oracle.init_state(n, seed)
    if three_d:
        rng = np.random.default_rng(seed)
        pos[:, 2] = rng.uniform(-100, 100, n).astype(np.float32)
        vel[:, 2] = rng.uniform(0, 0.1, n).astype(np.float32)
    return pos, vel


def fast_close(got_v, ref_v, vel0, tol=2e-5):
    """FAST's per-step bound: the velocity CHANGE within tol of its largest component (the reference's own sequential binary32
    sum is that far from the exact sum at these sizes) + the final rounding of v"""
    scale = float(np.abs(ref_v - vel0).max())
    ulp = float(np.spacing(np.float32(np.abs(ref_v).max())))
    err = float(np.abs(got_v - ref_v).max())
    assert err <= tol * scale + ulp, f"max |dv| {err:.3e} against {tol * scale + ulp:.3e} (scale {scale:.3e})"


# small sets, every shape of the kernel: np (bodies per lane / 2), a-blocks per launch, waves per a-block; even and odd numbers of
# blocks on the ring; 2, 3, 4 and 8 ranks; planar and 3-D data; shared and unshared reciprocal
@pytest.mark.parametrize("n,world,np_,ga,wpb", [
    (2048, 2, 4, 0, 0),      # two blocks of 512 per rank, NB = 4
    (1536, 3, 4, 0, 0),      # NB = 3: an odd ring, no antipodal block
    (4096, 4, 4, 1, 4),      # one a-block per launch: two launches per rank
    (4096, 8, 4, 0, 8),      # one block per rank, eight ranks
    (3072, 3, 2, 2, 0),      # blocks of 256, NB = 12, two a-blocks per launch of four
    (2560, 2, 2, 3, 12),     # NB = 10, a-blocks per launch that do not divide the rank's five
    (8192, 2, 4, 0, 0),
    (12288, 3, 4, 5, 0),
])
@pytest.mark.parametrize("three_d", [False, True], ids=["planar", "3d"])
def test_ring_shapes_vs_oracle(nb, oracle, monkeypatch, n, world, np_, ga, wpb, three_d):
    monkeypatch.setenv("NB_RING", "1")
    monkeypatch.setenv("NB_RING_NP", str(np_))
    if ga:
        monkeypatch.setenv("NB_RING_GA", str(ga))
    if wpb:
        monkeypatch.setenv("NB_RING_WPB", str(wpb))
    pos, vel = state(oracle, n, 100 + n + world, three_d)
    params = nb.default_params(mode=nb.NB_MODE_FAST)
    keep = {}
    p, v = ring_steps_on_one_gpu(nb, pos, vel, world, params, 1, keep)
    assert all(np.isfinite(s[:, :3]).all() and (s[:, 3] == 0).all() for s in keep["sums"]), "a record of `sums` was not written"
    p_ref, v_ref = oracle.run(pos, vel, 1)
    fast_close(v, v_ref, vel)
    assert np.abs(p - p_ref).max() <= np.abs(v - v_ref).max() + float(np.spacing(np.float32(np.abs(p_ref).max())))
    if not three_d:
        assert (p[:, 2] == 0).all() and (v[:, 2] == 0).all()
    # deterministic: a second run gives the same bits
    p2, v2 = ring_steps_on_one_gpu(nb, pos, vel, world, params, 1)
    assert (bits(p) == bits(p2)).all() and (bits(v) == bits(v2)).all()
    # three steps in a row (buffers reused, the launch generation moves on)
    p3, v3 = ring_steps_on_one_gpu(nb, pos, vel, world, params, 3)
    p_ref3, v_ref3 = oracle.run(pos, vel, 3)
    assert np.abs(p3 - p_ref3).max() <= 1e-4 and np.abs(v3 - v_ref3).max() <= 1e-5


# the step in PHASES (nb_launch_ring_fold_phase): the same decomposition, a round's worth of the pairs inside a rank's own slot apart
# from all others; every shape of the phases' kernels: blocks of 512 / 256, one block per rank (an own part of one block), odd and
# even rings, 2 / 3 / 4 / 8 ranks, sub-tiles per workgroup and the first phase's cap named and by default, planar and 3-D data
@pytest.mark.parametrize("n,world,np_,c4_own,c4_rest,cap", [
    (2048, 2, 4, 0, 0, 0),
    (1536, 3, 4, 0, 0, 0),      # an odd ring
    (4096, 8, 4, 0, 0, 0),      # one block per rank: its own part is the block against itself
    (4096, 4, 4, 4, 8, 4),      # the first phase takes half a block's inner pairs, the second the other half
    (3072, 3, 2, 4, 4, 8),      # blocks of 256 (four sub-tiles each)
    (8192, 2, 4, 8, 12, 24),    # two ranks: the second rank's first block has no pair with a rank in front
    (12288, 3, 4, 5, 7, 12),    # workgroup sizes that divide nothing (the cap is rounded to whole workgroups)
    (16384, 4, 4, 0, 0, 1000),  # a cap beyond every own part: all of it in the first phase
    (16384, 4, 4, 0, 0, 0),
])
@pytest.mark.parametrize("three_d", [False, True], ids=["planar", "3d"])
def test_ring_phases_vs_oracle(nb, oracle, monkeypatch, n, world, np_, c4_own, c4_rest, cap, three_d):
    monkeypatch.setenv("NB_RING", "1")
    monkeypatch.setenv("NB_RING_NP", str(np_))
    for name, v in (("NB_RING_C4_OWN", c4_own), ("NB_RING_C4_REST", c4_rest), ("NB_RING_CAP", cap)):
        if v:
            monkeypatch.setenv(name, str(v))
    pos, vel = state(oracle, n, 300 + n + world, three_d)
    params = nb.default_params(mode=nb.NB_MODE_FAST)
    keep = {}
    p, v = ring_steps_on_one_gpu(nb, pos, vel, world, params, 1, keep, phases=True)
    assert all(np.isfinite(s[:, :3]).all() and (s[:, 3] == 0).all() for s in keep["sums"]), "a record of `sums` was not written (or read a slot in flight)"
    p_ref, v_ref = oracle.run(pos, vel, 1)
    fast_close(v, v_ref, vel)
    assert np.abs(p - p_ref).max() <= np.abs(v - v_ref).max() + float(np.spacing(np.float32(np.abs(p_ref).max())))
    if not three_d:
        assert (p[:, 2] == 0).all() and (v[:, 2] == 0).all()
    # the one-launch form evaluates the same pairs: the two agree to the rounding of the sums' order
    p1, v1 = ring_steps_on_one_gpu(nb, pos, vel, world, params, 1)
    fast_close(v, v1, vel)
    # deterministic: a second run gives the same bits
    p2, v2 = ring_steps_on_one_gpu(nb, pos, vel, world, params, 1, phases=True)
    assert (bits(p) == bits(p2)).all() and (bits(v) == bits(v2)).all()
    # three steps in a row (buffers and flag words reused, the launch generation moves on)
    p3, v3 = ring_steps_on_one_gpu(nb, pos, vel, world, params, 3, phases=True)
    p_ref3, v_ref3 = oracle.run(pos, vel, 3)
    assert np.abs(p3 - p_ref3).max() <= 1e-4 and np.abs(v3 - v_ref3).max() <= 1e-5
    # the fused finish (the rank's own records added there, the next step's first phase on the planes it left): the same additions
    # in the same order -- the same bits
    p4, v4 = ring_steps_on_one_gpu(nb, pos, vel, world, params, 3, phases="fused")
    assert (bits(p3) == bits(p4)).all() and (bits(v3) == bits(v4)).all()


@pytest.mark.parametrize("cap", [0, 8])
@pytest.mark.parametrize("where", ["own", "front", "behind"])
def test_ring_phases_decide_their_arithmetic_from_what_they_read(nb, oracle, monkeypatch, where, cap):
    """Each phase has flag words of its own: the first phase's pairs are planar (or may share a reciprocal) when the rank's own slot
    is, whatever the rest of the set holds -- the rest has not arrived when it runs -- and the second phase obeys the whole set,
    also for the own-slot pairs the first left it (cap 8: most of them).  One rank's slot is made 3-D and gets a coordinate too
    large for the shared reciprocal; every rank's step must still be the oracle's."""
    monkeypatch.setenv("NB_RING", "1")
    if cap:
        monkeypatch.setenv("NB_RING_CAP", str(cap))
    n, world = 8192, 4
    S = n // world
    pos, vel = state(oracle, n, 77, False)
    slot = {"own": 0, "front": 1, "behind": 3}[where]          # seen from rank 0
    rng = np.random.default_rng(5)
    pos[slot * S:(slot + 1) * S, 2] = rng.uniform(-100, 100, S).astype(np.float32)
    pos[slot * S + 17, 0] = np.float32(3e8)
    params = nb.default_params(mode=nb.NB_MODE_FAST)
    _, v = ring_steps_on_one_gpu(nb, pos, vel, world, params, 1, phases=True)
    _, v_ref = oracle.run(pos, vel, 1)
    assert np.isfinite(v).all()
    fast_close(v, v_ref, vel)
    _, v2 = ring_steps_on_one_gpu(nb, pos, vel, world, params, 1, phases=True)
    assert (bits(v) == bits(v2)).all()
    # the fused finish stamps the NEXT step's verdicts for the own slot: three steps (a planar slot turns 3-D behind the first -- its
    # bodies were pulled out of the plane) give the bits of the launches that look at the records themselves
    p3, v3 = ring_steps_on_one_gpu(nb, pos, vel, world, params, 3, phases=True)
    p4, v4 = ring_steps_on_one_gpu(nb, pos, vel, world, params, 3, phases="fused")
    assert np.isfinite(v4).all() and (bits(p3) == bits(p4)).all() and (bits(v3) == bits(v4)).all()


def test_shapes_that_run_their_step_in_phases(nb, monkeypatch):
    from nenbody_amd.dist import HipBackend

    be = HipBackend()
    fast, strict = nb.default_params(mode=nb.NB_MODE_FAST), nb.default_params()
    for world in (2, 4, 8):   # BASELINE config 4 at every rank count
        assert be.ring_phased(fast, 131072, 0, 131072 // world) and be.ring_phased(fast, 131072, 131072 - 131072 // world, 131072 // world)
    assert be.ring_phased(fast, 65536, 0, 16384)
    assert not be.ring_phased(strict, 131072, 0, 16384)        # STRICT has no pairs form
    assert not be.ring_phased(fast, 131072, 0, 131072)
    assert not be.ring_phased(fast, 16384, 0, 2048)            # below the pairs form's own line
    # config 5: a rank's rows do not fit one launch (2 GB at 8 ranks), the step is walked in groups of a-blocks and stays
    # fold -> exchange -> finish -> all-gather: two exchanges of 2 MB per peer against 13.6 ms of compute
    for world in (2, 4, 8):
        assert be.ring_partners(fast, 1 << 20, 0, (1 << 20) // world) >= 1 and not be.ring_phased(fast, 1 << 20, 0, (1 << 20) // world)
    monkeypatch.setenv("NB_RING_GA", "8")                       # a-blocks per launch named: groups, no phases
    assert not be.ring_phased(fast, 131072, 0, 16384)


def test_ring_unshared_reciprocal_and_forced_3d(nb, oracle, monkeypatch):
    monkeypatch.setenv("NB_RING", "1")
    n, world = 4096, 4
    pos, vel = state(oracle, n, 9, False)
    params = nb.default_params(mode=nb.NB_MODE_FAST)
    _, v_ref = oracle.run(pos, vel, 1)
    monkeypatch.setenv("NB_FAST_NO_SHARE", "1")
    _, v = ring_steps_on_one_gpu(nb, pos, vel, world, params, 1)
    fast_close(v, v_ref, vel)
    monkeypatch.setenv("NB_FORCE_3D", "1")
    _, v = ring_steps_on_one_gpu(nb, pos, vel, world, params, 1)
    fast_close(v, v_ref, vel)
    # coordinates too large for the shared reciprocal (the product of two squared distances must stay normal): the flag word of
    # planes_kernel turns the sharing off for the step
    monkeypatch.delenv("NB_FAST_NO_SHARE")
    monkeypatch.delenv("NB_FORCE_3D")
    big = pos.copy()
    big[5, 0] = np.float32(3e8)
    _, v_ref = oracle.run(big, vel, 1)
    _, v = ring_steps_on_one_gpu(nb, big, vel, world, params, 1)
    fast_close(v, v_ref, vel)


def test_ring_coincident_bodies_and_the_self_pair(nb, oracle, monkeypatch):
    """a block swept against itself meets every body with itself: the self pair adds (0 * G) / bias = 0 as in the reference
    (main.rs:425 includes i == n), and coincident bodies add 0 to each other"""
    monkeypatch.setenv("NB_RING", "1")
    n, world = 2048, 2
    pos, vel = state(oracle, n, 31, False)
    pos[7] = pos[1500]          # a coincident pair across the ranks
    pos[100] = pos[101]         # and inside a block
    params = nb.default_params(mode=nb.NB_MODE_FAST)
    _, v = ring_steps_on_one_gpu(nb, pos, vel, world, params, 1)
    _, v_ref = oracle.run(pos, vel, 1)
    assert np.isfinite(v).all()
    fast_close(v, v_ref, vel)


def test_shapes_that_keep_the_ordered_fold(nb, monkeypatch):
    from nenbody_amd.dist import HipBackend

    be = HipBackend()
    fast, strict = nb.default_params(mode=nb.NB_MODE_FAST), nb.default_params()
    assert be.ring_partners(fast, 131072, 0, 16384) == 4 and be.ring_partners(fast, 131072, 16384 * 5, 16384) == 4
    assert be.ring_partners(fast, 131072, 0, 65536) == 1 and be.ring_partners(fast, 131072, 0, 32768) == 2
    assert be.ring_partners(fast, 1 << 20, 0, 1 << 17) == 4
    assert be.ring_partners(strict, 131072, 0, 16384) == 0          # STRICT keeps the reference's order of additions
    assert be.ring_partners(fast, 131072, 0, 131072) == 0           # a whole set has its own pairs form
    assert be.ring_partners(fast, 131072, 100, 16384) == 0          # not a rank of equal ranks
    assert be.ring_partners(fast, 131000, 0, 16375) == 0            # not whole blocks
    assert be.ring_partners(fast, 16384, 0, 2048) == 0              # small sets: the second exchange does not pay
    # the line is 2^30 ordered pairs per rank and step (profiles/r04/ring_small.log)
    assert be.ring_partners(fast, 65536, 0, 16384) == 2 and be.ring_partners(fast, 65536, 0, 8192) == 0
    assert be.ring_partners(fast, 32768, 0, 16384) == 0 and be.ring_partners(fast, 65536, 0, 32768) == 1
    monkeypatch.setenv("NB_RING", "0")
    assert be.ring_partners(fast, 131072, 0, 16384) == 0
    assert be.ring_scratch_bytes(fast, 131072, 0, 16384) == 0


# BASELINE configs 4 and 5 in the pairs form: every rank's launch set at 2, 4 and 8 ranks
@pytest.mark.parametrize("n,world,phases", [(131072, 2, False), (131072, 4, False), (131072, 8, False), (1 << 20, 2, False), (1 << 20, 4, False),
                                            (1 << 20, 8, False), (131072, 2, True), (131072, 4, True), (131072, 8, True)], ids=lambda x: str(x))
def test_every_rank_of_configs_4_and_5_in_the_pairs_form(nb, oracle, n, world, phases):
    pos, vel = nb.init_state(n, 1234)
    params = nb.default_params(mode=nb.NB_MODE_FAST)
    p, v = ring_steps_on_one_gpu(nb, pos, vel, world, params, 1, phases=phases)
    S = n // world
    # first, last and six inner bodies of every rank against the oracle and against the same sum carried in binary64
    idx = np.unique(np.concatenate([np.concatenate([[r * S, r * S + S - 1], np.linspace(r * S + 1, r * S + S - 2, 6).astype(np.int64)])
                                    for r in range(world)]))
    c = [float(np.float32(x)) for x in (0.1, 0.001, 0.0000001)]
    from test_gpu_parity import _oracle_bodies

    dv64, v_ref = _oracle_bodies(oracle, pos, vel, idx, c)
    v_true = vel[idx].astype(np.float64) + dv64
    scale = np.abs(dv64).max()
    ulp_v = float(np.spacing(np.float32(np.abs(v_true).max())))
    err_ref = np.abs(v_ref.astype(np.float64) - v_true).max(axis=1)
    err_fast = np.abs(v[idx].astype(np.float64) - v_true).max(axis=1)
    msg = (f"n={n} ranks={world}: max |v - v64| / max|dv|: reference binary32 {err_ref.max() / scale:.2e}, pairs form on shards "
           f"{err_fast.max() / scale:.2e}; max |FAST - reference| / max|dv| {np.abs(v[idx] - v_ref).max() / scale:.2e}")
    print(msg)
    assert err_fast.max() <= err_ref.max() + ulp_v, msg           # no further from the exact sum than the reference's own arithmetic
    assert (err_fast <= err_ref + 2e-5 * scale + ulp_v).all(), msg
    tol = 2e-5 if n <= 131072 else 1e-4
    assert np.abs(v[idx] - v_ref).max() <= tol * scale + ulp_v, msg
    # all bodies against the one-GPU FAST step (itself held to the oracle in test_gpu_parity): the bulk at FAST's tolerance, the
    # worst body (a neighbour at r ~ 1e-4 puts one term ~ 1 into its sum) at the looser bound used there
    with nb.Scene(pos, vel, params) as sc:
        sc.step_n(1)
        p1, v1 = sc.state()
    dv_all = np.abs(v - v1).max(axis=1)
    assert np.quantile(dv_all, 0.999) <= 2 * tol * scale + ulp_v and dv_all.max() <= 1e-3 * scale, msg
    assert np.abs(p - p1).max() <= dv_all.max() + float(np.spacing(np.float32(np.abs(p1).max()))), msg
    # run to run: the same bits
    if n <= 131072:
        p2, v2 = ring_steps_on_one_gpu(nb, pos, vel, world, params, 1, phases=phases)
        assert (bits(p) == bits(p2)).all() and (bits(v) == bits(v2)).all()


def test_ring_form_tracks_the_reference_as_the_one_gpu_fast_step_does(nb, capsys):
    """ten steps of the headline set on 8 ranks in the pairs form, against STRICT (= the reference, bit for bit) and against the
    one-GPU FAST step: the typical body stays inside the north_star's 1e-4 through the free fall, the worst one leaves it at step 2
    exactly as FAST's does on one GPU (the closest pair: tests/test_gpu_parity.py::test_headline_size_fast_drift_curve) -- sharding
    the pairs changes which GPU evaluates them, not how well"""
    n, world, k = 131072, 8, 10
    pos, vel = nb.init_state(n, 1234)
    fast = nb.default_params(mode=nb.NB_MODE_FAST)
    with nb.Scene(pos, vel) as ref, nb.Scene(pos, vel, fast) as one:
        ref.step_n(k)
        one.step_n(k)
        (pr, _), (p1, _) = ref.state(), one.state()
    p8, _ = ring_steps_on_one_gpu(nb, pos, vel, world, fast, k)
    d8 = np.abs(p8.astype(np.float64) - pr).max(axis=1)
    d1 = np.abs(p1.astype(np.float64) - pr).max(axis=1)
    with capsys.disabled():
        print(f"\n  |dr| against STRICT after {k} steps at N = {n} (max / 99.9 % / median): 8 ranks in the pairs form {d8.max():.1e} / "
              f"{np.quantile(d8, 0.999):.1e} / {np.median(d8):.1e}; one GPU {d1.max():.1e} / {np.quantile(d1, 0.999):.1e} / {np.median(d1):.1e}")
    assert np.median(d8) < 1e-4 and np.quantile(d8, 0.999) < 2e-3
    assert np.median(d8) <= 2 * np.median(d1) + 1e-5 and np.quantile(d8, 0.999) <= 3 * np.quantile(d1, 0.999) + 1e-5


def test_ring_form_is_deterministic_and_finite_through_the_collapse(nb):
    """100 steps of the headline set on 8 ranks in the pairs form, twice: the same bits (no sum depends on which workgroup or
    which rank finished first: rows in a-block order, quarters in order, received chunks in ascending distance), and no body goes
    non-finite while the cloud collapses and rebounds (step ~40)"""
    n, world = 131072, 8
    pos, vel = nb.init_state(n, 1234)
    fast = nb.default_params(mode=nb.NB_MODE_FAST)
    pa, va = ring_steps_on_one_gpu(nb, pos, vel, world, fast, 100)
    pb, vb = ring_steps_on_one_gpu(nb, pos, vel, world, fast, 100)
    assert (bits(pa) == bits(pb)).all() and (bits(va) == bits(vb)).all()
    assert np.isfinite(pa).all() and np.isfinite(va).all()
    assert (pa[:, 2] == 0).all()
    # the same in phases (the exchanges behind compute): its own bits, as reproducible, as finite
    pc, vc = ring_steps_on_one_gpu(nb, pos, vel, world, fast, 100, phases=True)
    pd, vd = ring_steps_on_one_gpu(nb, pos, vel, world, fast, 100, phases="fused")   # ... and with the fused finish: the same bits
    assert (bits(pc) == bits(pd)).all() and (bits(vc) == bits(vd)).all()
    assert np.isfinite(pc).all() and np.isfinite(vc).all() and (pc[:, 2] == 0).all()
