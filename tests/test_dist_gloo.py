"""world_size-2 (and 3) runs of the sharded scene over gloo on CPU: the N > 1 orchestration --
index-range partition, local update, one all-gather of positions per step, buffer ping-pong -- must give the
same bits as the unsharded run, for even and ragged splits."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, k, out_dir, boids=False, overlap=False, ring=False, slow=None, ring_overlap=None):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import nenbody_amd
        import oracle
        from oracle_backend import OracleBackend

        pos, vel = oracle.init_state(n, seed=4321)
        pos[:, 2] = np.linspace(-1, 1, n, dtype=np.float32)
        params = nenbody_amd.default_params(mode=nenbody_amd.NB_MODE_FAST) if (overlap or ring) else None
        sc = nenbody_amd.ShardedScene(pos, vel, params, backend=OracleBackend(), device="cpu", overlap=overlap, ring=True if ring else False,
                                      ring_overlap=ring_overlap)
        assert (sc.first, sc.count) == nenbody_amd.partition(n, world)[rank] and sc.overlap == overlap and sc.ring_overlap == bool(ring_overlap)
        if ring and slow:   # choose_form: both forms timed on the state in hand, the slower rank's time decides, the state is put back
            import time as _time

            sc.step()
            before = (sc.positions().copy(), sc.velocities().copy(), sc.cur, sc.steps_done)
            names = ("ring_fold", "ring_fold_phase") if slow == "pairs" else ("step",)
            reals = {name: getattr(sc.backend, name) for name in names}

            def slowed(real):
                def call(*a, **kw):
                    if rank == world - 1:   # one slow rank is enough: every rank takes the slowest rank's time
                        _time.sleep(0.05)
                    return real(*a, **kw)
                return call

            for name in names:
                setattr(sc.backend, name, slowed(reals[name]))
            chosen = sc.choose_form(steps=2, warm=1)
            for name in names:
                setattr(sc.backend, name, reals[name])
            # three candidates: the pairs form with its exchanges in sequence / behind compute, and the ordered fold
            assert set(sc.form_times) == {"pairs", "pairs_overlapped", "ordered"}
            assert (chosen == "ordered") if slow == "pairs" else chosen.startswith("pairs"), (chosen, sc.form_times)
            assert sc.partners == (0 if chosen == "ordered" else ring)
            assert sc.ring_overlap == (chosen == "pairs_overlapped")
            assert (sc.positions() == before[0]).all() and (sc.velocities() == before[1]).all() and (sc.cur, sc.steps_done) == before[2:]
            sc.step_n(k - 1)
        elif ring:   # two exchanges per step: the halves that belong to the ranks in front (point to point), then the positions
            assert sc.partners == ring and not sc.overlap
            sent, gathers, order = [], [], []
            real_batch, real_gather = dist.batch_isend_irecv, dist.all_gather_into_tensor

            def counting_batch(ops):
                sent.append(sorted((op.op.__name__, op.peer, op.tensor.numel()) for op in ops))
                order.append("exchange")
                return real_batch(ops)

            def counting_gather(out, inp, *a, **kw):
                gathers.append(out.numel())
                order.append("gather")
                return real_gather(out, inp, *a, **kw)

            def logged(name, real):   # the launches between the collectives, in the order they are issued
                def call(*a, **kw):
                    order.append(name if name != "ring_fold_phase" else f"phase{a[4]}")
                    return real(*a, **kw)
                return call

            real_finish = sc.backend.ring_finish   # (the mirror's fused finish calls it: count the fused one only)
            for name in ("ring_fold", "ring_fold_phase", "ring_finish", "ring_finish_phase"):
                setattr(sc.backend, name, logged(name, getattr(sc.backend, name)))
            if ring_overlap:
                sc.backend.ring_finish = real_finish
            dist.batch_isend_irecv, dist.all_gather_into_tensor = counting_batch, counting_gather
            sc.step_n(k)
            dist.batch_isend_irecv, dist.all_gather_into_tensor = real_batch, real_gather
            want = sorted([("isend", (rank + d) % world, sc.count * 4) for d in range(1, ring + 1)] +
                          [("irecv", (rank - d) % world, sc.count * 4) for d in range(1, ring + 1)])
            assert sent == [want] * k and gathers == [world * sc.slot * 4] * k, (sent, gathers)
            # one step: pairs that need no other GPU first (phase 1; phase 4 from the second step on: the fused finish left their
            # planes), every other pair (2), the second exchange as soon as the sums of the ranks in front exist, the rank's own sums (3)
            # beside it, the fused finish, the all-gather -- which the NEXT step's first phase does not wait for
            per_step = (["phase4", "phase2", "exchange", "phase3", "ring_finish_phase", "gather"] if ring_overlap
                        else ["ring_fold", "exchange", "ring_finish", "gather"])
            want_order = per_step * k
            if ring_overlap:
                want_order[0] = "phase1"
            assert order == want_order, order
        elif boids:   # boids, n-body, boids: the velocity replica must be rebuilt after the n-body step
            gathers = []
            real = dist.all_gather_into_tensor

            def counting(out, inp, *a, **kw):
                gathers.append(out.numel())
                return real(out, inp, *a, **kw)

            dist.all_gather_into_tensor = counting
            sc.step_boids()
            sc.step()
            for _ in range(k):
                sc.step_boids()
            dist.all_gather_into_tensor = real
            # ONE exchange per boids step (positions and velocities in one staging buffer: twice a replica's size), one per n-body
            # step, and one rebuild of the velocity replica each time boids follows something else
            slots = world * sc.slot * 4
            assert gathers == [slots, 2 * slots, slots, slots] + [2 * slots] * k, gathers
        else:
            sc.step_n(k)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), pos=sc.positions(), vel=sc.velocities(),
                 inst=sc.local_instances(), first=sc.first, count=sc.count)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,k", [(2, 64, 5), (2, 37, 4), (3, 50, 3), (8, 70, 2)])
def test_sharded_equals_unsharded(tmp_path, oracle, world, n, k):
    mp.spawn(_worker, args=(world, _free_port(), n, k, str(tmp_path)), nprocs=world, join=True)
    pos, vel = oracle.init_state(n, seed=4321)
    pos[:, 2] = np.linspace(-1, 1, n, dtype=np.float32)
    p_ref, v_ref, inst_ref = oracle.run(pos, vel, k, want_instances=True)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        # every rank holds the full position replica and can gather all velocities
        assert (got["pos"].view(np.uint32) == p_ref.view(np.uint32)).all()
        assert (got["vel"].view(np.uint32) == v_ref.view(np.uint32)).all()
        f, c = int(got["first"]), int(got["count"])
        assert (got["inst"].view(np.uint32) == inst_ref[f:f + c].view(np.uint32)).all()


def test_world_of_one_needs_no_process_group(oracle):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import nenbody_amd
    from oracle_backend import OracleBackend

    pos, vel = oracle.init_state(20, seed=1)
    sc = nenbody_amd.ShardedScene(pos, vel, backend=OracleBackend(), device="cpu")
    sc.step_n(3)
    p_ref, v_ref = oracle.run(pos, vel, 3)
    assert (sc.positions().view(np.uint32) == p_ref.view(np.uint32)).all()
    assert (sc.velocities().view(np.uint32) == v_ref.view(np.uint32)).all()


@pytest.mark.parametrize("world,n,k", [(2, 60, 2), (3, 41, 2)])
def test_sharded_boids_equals_unsharded(tmp_path, oracle, world, n, k):
    """Boids gathers positions AND velocities every step (main.rs:494-504 reads every old velocity)."""
    mp.spawn(_worker, args=(world, _free_port(), n, k, str(tmp_path), True), nprocs=world, join=True)
    pos, vel = oracle.init_state(n, seed=4321)
    pos[:, 2] = np.linspace(-1, 1, n, dtype=np.float32)
    p_ref, v_ref = oracle.boids_run(pos, vel, 1)
    p_ref, v_ref = oracle.run(p_ref, v_ref, 1)
    p_ref, v_ref = oracle.boids_run(p_ref, v_ref, k)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        assert (got["pos"].view(np.uint32) == p_ref.view(np.uint32)).all()
        assert (got["vel"].view(np.uint32) == v_ref.view(np.uint32)).all()


@pytest.mark.parametrize("world,n,k", [(2, 64, 5), (3, 50, 4), (3, 2, 3)])
def test_sharded_fast_with_overlapped_exchange(tmp_path, oracle, world, n, k):
    """overlap=True (FAST): each step folds this rank's own slot of the new snapshot first, then waits for the exchange of the
    others, folds the rest and integrates (nb_launch_step_phase).  Every record of the snapshot must have been read after
    it arrived: the run must match the unsharded run to FAST's tolerance on every rank (ranks with no bodies included).
    With STRICT the flag is ignored: the reference's order of additions stays."""
    mp.spawn(_worker, args=(world, _free_port(), n, k, str(tmp_path), False, True), nprocs=world, join=True)
    pos, vel = oracle.init_state(n, seed=4321)
    pos[:, 2] = np.linspace(-1, 1, n, dtype=np.float32)
    p_ref, v_ref = oracle.run(pos, vel, k)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        assert np.abs(got["pos"] - p_ref).max() <= 2e-5 and np.abs(got["vel"] - v_ref).max() <= 1e-6
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import nenbody_amd
    from oracle_backend import OracleBackend

    sc = nenbody_amd.ShardedScene(pos, vel, backend=OracleBackend(), device="cpu", overlap=True)   # STRICT, world 1
    assert not sc.overlap


@pytest.mark.parametrize("world,n,k,partners", [(2, 64, 4, 1), (3, 48, 3, 2), (4, 64, 3, 2), (8, 64, 2, 4)])
def test_sharded_fast_pairs_with_the_exchanges_behind_compute(tmp_path, oracle, world, n, k, partners, ring_overlap=True):
    """ring_overlap (FAST, equal ranks; nb_launch_ring_fold_phase): pairs inside a rank's own slot need no other GPU, so a round's
    worth of them runs while the last step's all-gather is still landing; the second exchange leaves as soon as the sums of the
    ranks in front are final and the rank's own sums are made beside it.  The worker checks the order of launches and collectives
    of every step and the peers and sizes of every message; the test double computes the first phase from a snapshot whose other
    slots are NaN (a record read before it could have arrived poisons the run); here: every rank's replica matches the unsharded
    run to FAST's tolerance, on worlds of 2, 3, 4 and 8."""
    mp.spawn(_worker, args=(world, _free_port(), n, k, str(tmp_path), False, False, partners, None, ring_overlap), nprocs=world, join=True)
    pos, vel = oracle.init_state(n, seed=4321)
    pos[:, 2] = np.linspace(-1, 1, n, dtype=np.float32)
    p_ref, v_ref = oracle.run(pos, vel, k)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        assert np.isfinite(got["pos"]).all() and np.isfinite(got["vel"]).all()
        assert np.abs(got["pos"] - p_ref).max() <= 2e-5 and np.abs(got["vel"] - v_ref).max() <= 1e-6


@pytest.mark.parametrize("world,n,k,partners", [(2, 64, 4, 1), (3, 48, 3, 2), (3, 51, 3, 2), (4, 64, 3, 2), (8, 64, 2, 4), (5, 55, 2, 3)])
def test_sharded_fast_pairs_once_with_second_exchange(tmp_path, oracle, world, n, k, partners):
    """ring=True (FAST, equal ranks): every unordered pair is evaluated once, by the rank that owns the body further back on the
    ring of indices; the other body's half travels to its owner in a second, point-to-point exchange (nb_launch_ring_fold /
    nb_launch_ring_finish).  The worker checks the collectives of every step (D sends to the ranks in front, D receives from the
    ranks behind, one all-gather); here: every rank's replica matches the unsharded run to FAST's tolerance."""
    mp.spawn(_worker, args=(world, _free_port(), n, k, str(tmp_path), False, False, partners), nprocs=world, join=True)
    pos, vel = oracle.init_state(n, seed=4321)
    pos[:, 2] = np.linspace(-1, 1, n, dtype=np.float32)
    p_ref, v_ref = oracle.run(pos, vel, k)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        assert np.abs(got["pos"] - p_ref).max() <= 2e-5 and np.abs(got["vel"] - v_ref).max() <= 1e-6


def test_ring_needs_equal_ranks(oracle):
    """a ragged split (or STRICT) keeps the ordered fold and its one exchange; asking for the ring there is an error"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import nenbody_amd
    from oracle_backend import OracleBackend

    pos, vel = oracle.init_state(50, seed=1)
    fast = nenbody_amd.default_params(mode=nenbody_amd.NB_MODE_FAST)
    sc = nenbody_amd.ShardedScene(pos, vel, fast, backend=OracleBackend(), device="cpu", rank=1, world=3)   # 17 + 17 + 16
    assert sc.partners == 0
    with pytest.raises(ValueError):
        nenbody_amd.ShardedScene(pos, vel, fast, backend=OracleBackend(), device="cpu", rank=1, world=3, ring=True)
    sc = nenbody_amd.ShardedScene(pos[:48], vel[:48], backend=OracleBackend(), device="cpu", rank=1, world=3)   # STRICT
    assert sc.partners == 0


@pytest.mark.parametrize("world,n,k,partners,slow", [(2, 64, 3, 1, "pairs"), (3, 48, 3, 2, "ordered")])
def test_choose_form_times_both_forms_and_every_rank_agrees(tmp_path, oracle, world, n, k, partners, slow):
    """ShardedScene.choose_form: where the pairs form is planned, both forms are timed on the current state (the slowest rank's
    time counts), the faster one is kept and the state is put back -- here one form is slowed down on ONE rank, so every rank must
    pick the other; stepping on gives the unsharded result either way."""
    mp.spawn(_worker, args=(world, _free_port(), n, k, str(tmp_path), False, False, partners, slow), nprocs=world, join=True)
    pos, vel = oracle.init_state(n, seed=4321)
    pos[:, 2] = np.linspace(-1, 1, n, dtype=np.float32)
    p_ref, v_ref = oracle.run(pos, vel, k)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        assert np.abs(got["pos"] - p_ref).max() <= 2e-5 and np.abs(got["vel"] - v_ref).max() <= 1e-6


def _verify_worker(rank, world, port, n, k, out_dir, broken):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import nenbody_amd
        import oracle
        from oracle_backend import OracleBackend

        pos, vel = oracle.init_state(n, seed=4321)
        pos[:, 2] = np.linspace(-1, 1, n, dtype=np.float32)
        sc = nenbody_amd.ShardedScene(pos, vel, nenbody_amd.default_params(mode=nenbody_amd.NB_MODE_FAST), backend=OracleBackend(), device="cpu",
                                      ring=True, ring_overlap=True)
        real_batch = dist.batch_isend_irecv

        def faulty(ops):   # a second exchange that loses data: what arrives is overwritten (one rank is enough to move every rank)
            reqs = real_batch(ops)
            if (broken == "grouped" and len(ops) > 2) or broken == "all":
                for r in reqs:
                    r.wait()
                if rank == 0:
                    for op in ops:
                        if op.op is dist.irecv:
                            op.tensor.add_(1.0)
                return []    # (all waited for: a gloo receive must not be waited for twice)
            return reqs

        if broken:
            dist.batch_isend_irecv = faulty
        rep = sc.verify_exchanges()
        dist.batch_isend_irecv = real_batch if broken != "all" else faulty
        want = {None: "grouped", "grouped": "per_distance", "all": "disabled"}[broken]
        if world == 2 and broken == "grouped":
            want = "grouped"    # (one partner: the group IS one send and one receive -- nothing to fall back from)
        assert rep["verified"] and rep["all_gather"] == "in_place" and rep["ring_exchange"] == want, rep
        assert sc.exchange_report is rep and (sc.partners == 0) == (want == "disabled") and sc.ring_grouped == (want == "grouped")
        sc.step_n(k)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), pos=sc.positions(), vel=sc.velocities())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,broken", [(3, 48, None), (4, 64, "grouped"), (3, 48, "all"), (8, 64, "grouped")])
def test_exchanges_are_verified_on_a_pattern_and_fall_back(tmp_path, oracle, world, n, broken):
    """ShardedScene.verify_exchanges: before the first step both exchanges move a known per-rank pattern and every rank checks what
    arrived (one all-reduce per check: all ranks agree on the verdict).  A second exchange that loses data when issued as one group
    moves to one group per distance; one that loses data either way is dropped for the ordered fold and its one exchange -- and the
    steps that follow are right in every case."""
    k = 2
    mp.spawn(_verify_worker, args=(world, _free_port(), n, k, str(tmp_path), broken), nprocs=world, join=True)
    pos, vel = oracle.init_state(n, seed=4321)
    pos[:, 2] = np.linspace(-1, 1, n, dtype=np.float32)
    p_ref, v_ref = oracle.run(pos, vel, k)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        assert np.abs(got["pos"] - p_ref).max() <= 2e-5 and np.abs(got["vel"] - v_ref).max() <= 1e-6
