"""CPU tests of the C-ABI shared library: it loads, exports exactly what include/nenbody.h declares,
validates arguments on the host, and -- with no GPU -- refuses to compute instead of falling back."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "nenbody.h")
DIAG_HEADER = os.path.join(ROOT, "include", "nenbody_diag.h")


def declared_functions(header=HEADER):
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nb_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_declare_the_same_symbols(nb):
    from nenbody_amd import _lib

    assert declared_functions() == sorted(_lib.PROTOTYPES)
    assert declared_functions(DIAG_HEADER) == sorted(_lib.DIAG_PROTOTYPES)
    # the boundary header declares nothing diagnostic: a host binds only what stands in for the reference's interface
    assert not [f for f in declared_functions() if f.startswith(("nb_selftest", "nb_debug", "nb_diag"))]


def test_library_exports_every_declared_symbol(nb):
    lib = ctypes.CDLL(nb._lib.LIB_PATH)
    for name in declared_functions() + declared_functions(DIAG_HEADER):
        assert hasattr(lib, name), f"libnenbody_hip.so does not export {name}"


def test_abi_version_and_default_params(nb):
    from nenbody_amd import _lib

    assert _lib.load().nb_abi_version() == _lib.NB_ABI_VERSION
    text = open(HEADER).read()
    assert int(re.search(r"#define NB_ABI_VERSION (\d+)", text).group(1)) == _lib.NB_ABI_VERSION
    p = nb.default_params()
    # src/main.rs:411-413
    assert p.dt == np.float32(0.1) and p.G == np.float32(0.001) and p.bias == np.float32(0.0000001)
    assert p.tile == 0 and p.mode == nb.NB_MODE_STRICT


def test_init_state_matches_the_oracles_generator(nb, oracle):
    for n, seed in [(1, 0), (33, 1234), (1000, 7)]:
        p, v = nb.init_state(n, seed)
        po, vo = oracle.init_state(n, seed)
        assert (p.view(np.uint32) == po.view(np.uint32)).all() and (v.view(np.uint32) == vo.view(np.uint32)).all()


def _create(nb, n, ndev, params):
    from nenbody_amd import _lib

    ctx = ctypes.c_void_p()
    rc = _lib.load().nb_create(n, ndev, ctypes.byref(params) if params is not None else None, ctypes.byref(ctx))
    return rc, ctx


def test_create_rejects_bad_arguments_before_touching_the_device(nb):
    from nenbody_amd import _lib

    p = nb.default_params()
    rc, _ = _create(nb, 0, 1, p)
    assert rc == _lib.NB_ERR_INVALID and "count" in _lib.last_error()
    rc, _ = _create(nb, 16, 2, p)
    assert rc == _lib.NB_ERR_UNSUPPORTED and "one process" in _lib.last_error()
    bad = nb.default_params()
    bad.mode = 7
    rc, _ = _create(nb, 16, 1, bad)
    assert rc == _lib.NB_ERR_INVALID and "mode" in _lib.last_error()
    bad = nb.default_params()
    bad.tile = 100
    rc, _ = _create(nb, 16, 1, bad)
    assert rc == _lib.NB_ERR_INVALID and "tile" in _lib.last_error()
    assert _lib.load().nb_create(16, 1, None, None) == _lib.NB_ERR_INVALID


def test_launch_step_validates_pointers_and_ranges(nb):
    from nenbody_amd import _lib

    lib = _lib.load()
    p = nb.default_params()
    fake_a, fake_b, fake_v = 0x1000, 0x2000, 0x3000   # never dereferenced: validation fails first
    assert lib.nb_launch_step(ctypes.byref(p), 16, 0, 16, None, fake_b, fake_v, None, 0, None) == _lib.NB_ERR_INVALID
    assert lib.nb_launch_step(ctypes.byref(p), 16, 0, 16, fake_a, fake_a, fake_v, None, 0, None) == _lib.NB_ERR_INVALID
    assert "alias" in _lib.last_error()
    assert lib.nb_launch_step(ctypes.byref(p), 16, 8, 9, fake_a, fake_b, fake_v, None, 0, None) == _lib.NB_ERR_INVALID
    assert "exceeds" in _lib.last_error()
    assert lib.nb_launch_step(ctypes.byref(p), 16, 0, 0, fake_a, fake_b, fake_v, None, 0, None) == _lib.NB_ERR_INVALID
    fast = nb.default_params(mode=nb.NB_MODE_FAST)
    need = lib.nb_scratch_bytes(ctypes.byref(fast), 131072, 16384)
    assert need > 0   # small shard of a large set: the j range is split, partial sums need scratch
    assert lib.nb_launch_step(ctypes.byref(fast), 131072, 0, 16384, fake_a, fake_b, fake_v, None, 0, None) == _lib.NB_ERR_INVALID
    assert "scratch" in _lib.last_error()


def test_scratch_bytes_host_arithmetic(nb):
    from nenbody_amd import _lib

    lib = _lib.load()
    strict = nb.default_params()
    # STRICT never splits the fold, but both of its large-set forms read x / y / z planes of the whole set (+ flags): the
    # scalar-load form of whole sets and the block chain of small shards
    assert lib.nb_scratch_bytes(ctypes.byref(strict), 131072, 131072) == 256 + 3 * 4 * 131072 + 64
    assert lib.nb_scratch_bytes(ctypes.byref(strict), 131072, 16384) == 256 + 3 * 4 * 131072 + 64
    tiled = nb.default_params(tile=1024)
    assert lib.nb_scratch_bytes(ctypes.byref(tiled), 131072, 131072) == 0     # naming an LDS tile asks for the LDS-tiled kernel
    assert lib.nb_scratch_bytes(ctypes.byref(strict), 1000, 1000) == 0          # small sets: producer/consumer, no scratch
    fast = nb.default_params(mode=nb.NB_MODE_FAST)
    # FAST at this size reads the planes too (scalar-load form), followed by the rows of partial sums of its grid.y slices
    b = lib.nb_scratch_bytes(ctypes.byref(fast), 131072, 16384) - (256 + 3 * 4 * 131072 + 64)
    assert b % (16384 * 16) == 0 and 2 <= b // (16384 * 16) <= 64
    assert lib.nb_scratch_bytes(ctypes.byref(fast), 10, 20) == 0             # invalid shape -> 0


def test_no_device_means_no_compute(nb):
    """The product has no CPU path: without a GPU the context cannot even be created."""
    from nenbody_amd import _lib

    if _lib.load().nb_device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(nb.NbError) as ei:
        nb.Scene.new(64)
    assert ei.value.status == _lib.NB_ERR_NO_DEVICE and "no CPU path" in str(ei.value)
    pos, vel = nb.init_state(8)
    inst = np.zeros((8, 4, 4), np.float32)
    with pytest.raises(nb.NbError):
        nb.update_instance_nbody(inst, pos, pos.copy(), vel, vel.copy())
    with pytest.raises(nb.NbError):
        nb.ShardedScene(pos, vel)


def test_update_instance_nbody_argument_contract(nb):
    """Errors the reference raises before computing anything (copy_from_slice panics on a length mismatch, main.rs:415-416)."""
    pos, vel = nb.init_state(8)
    inst = np.zeros((8, 4, 4), np.float32)
    with pytest.raises(ValueError, match="old_positions"):
        nb.update_instance_nbody(inst, pos, np.zeros((7, 3), np.float32), vel, vel.copy())
    with pytest.raises(ValueError, match="old_velocities"):
        nb.update_instance_nbody(inst, pos, pos.copy(), vel, np.zeros((9, 3), np.float32))
    with pytest.raises(TypeError):
        nb.update_instance_nbody(inst, pos.astype(np.float64), pos.copy(), vel, vel.copy())
    # an empty zip updates nothing but still takes the snapshot copies (main.rs:415-416 run first)
    old_p, old_v = np.zeros_like(pos), np.zeros_like(vel)
    nb.update_instance_nbody(np.zeros((0, 4, 4), np.float32), pos, old_p, vel, old_v)
    assert (old_p == pos).all() and (old_v == vel).all()


def test_context_calls_reject_a_null_context(nb):
    from nenbody_amd import _lib

    lib = _lib.load()
    assert lib.nb_step_boids(None, 1, None) == _lib.NB_ERR_INVALID
    assert lib.nb_step_random(None, 1, 0) == _lib.NB_ERR_INVALID
    assert lib.nb_device_state(None, None, None, None) == _lib.NB_ERR_INVALID
    assert lib.nb_cameras(None, None, None, None) == _lib.NB_ERR_INVALID
    assert lib.nb_steps_done(None) == 0


def test_boids_defaults_and_launch_validation(nb):
    from nenbody_amd import _lib

    lib = _lib.load()
    bp = nb.default_boids_params()
    # src/main.rs:450-456
    assert (bp.dt, bp.rule_1_distance, bp.rule_2_distance, bp.rule_3_distance) == (np.float32(0.04), 1000.0, 5.0, 500.0)
    assert (bp.rule_1_scale, bp.rule_2_scale, bp.rule_3_scale) == (np.float32(0.02), np.float32(0.05), 0.5)
    a, b, c, d = 0x1000, 0x2000, 0x3000, 0x4000   # never dereferenced: validation fails first
    assert lib.nb_launch_boids_step(ctypes.byref(bp), 16, 0, 16, a, b, a, d, None) == _lib.NB_ERR_INVALID   # pos aliases
    assert lib.nb_launch_boids_step(ctypes.byref(bp), 16, 0, 16, a, b, c, b, None) == _lib.NB_ERR_INVALID   # vel aliases
    assert lib.nb_launch_boids_step(ctypes.byref(bp), 16, 8, 9, a, b, c, d, None) == _lib.NB_ERR_INVALID    # range
    bp.tile = 100
    assert lib.nb_launch_boids_step(ctypes.byref(bp), 16, 0, 16, a, b, c, d, None) == _lib.NB_ERR_INVALID
    assert "tile" in _lib.last_error()
    assert lib.nb_launch_cameras(0, a, b, c, d, a, None) == _lib.NB_ERR_INVALID
    assert lib.nb_launch_random_step(0, 0, a, b, 1, 0, None) == _lib.NB_ERR_INVALID


def test_boids_split_scratch_for_a_rank_without_bodies(nb):
    """sizing the split form's scratch for an empty rank (count == 0, as every other shard call accepts) divided by
    zero and killed the host with SIGFPE.  It needs no scratch: 0; a rank with bodies gets rows for every slice."""
    import ctypes

    lib = nb.load()
    bp = nb._lib.default_boids_params()
    assert lib.nb_boids_split_scratch_bytes(ctypes.byref(bp), 131072, 0) == 0
    assert lib.nb_boids_split_scratch_bytes(None, 1000, 0) == 0
    some = lib.nb_boids_split_scratch_bytes(ctypes.byref(bp), 131072, 16384)
    assert some >= 16384 * 3 * 16      # at least one slice of rows
    assert lib.nb_boids_split_scratch_bytes(ctypes.byref(bp), 131072, 1) > 0


def test_shard_argument_contract_and_no_device(nb):
    """nb_shard_*: arguments are validated before any device work; without a GPU nothing can be created."""
    from nenbody_amd import _lib

    lib = _lib.load()
    sh = ctypes.c_void_p()
    for n, rank, world in ((0, 0, 1), (8, 0, 0), (8, 2, 2), (8, -1, 2)):
        assert lib.nb_shard_create(n, rank, world, None, ctypes.byref(sh)) == _lib.NB_ERR_INVALID
        assert not sh.value and b"rank" in lib.nb_last_error(None)
    assert lib.nb_shard_create(8, 0, 1, None, None) == _lib.NB_ERR_INVALID
    bad = _lib.default_params()
    bad.mode = 9
    assert lib.nb_shard_create(8, 0, 2, ctypes.byref(bad), ctypes.byref(sh)) == _lib.NB_ERR_INVALID
    for fn, args in ((lib.nb_shard_step, (None, 1)), (lib.nb_shard_step_boids, (None, 1, None)), (lib.nb_shard_sync, (None,)),
                     (lib.nb_shard_upload, (None, None, None)), (lib.nb_shard_download, (None, None, None, None)),
                     (lib.nb_shard_range, (None, None, None)), (lib.nb_shard_use_rccl, (None, None)),
                     (lib.nb_comm_id, (None,))):
        assert fn(*args) == _lib.NB_ERR_INVALID
    lib.nb_shard_destroy(None)
    assert lib.nb_shard_last_error(None) is not None
    if lib.nb_device_count() > 0:
        pytest.skip("a HIP device is present")
    assert lib.nb_shard_create(8, 0, 2, None, ctypes.byref(sh)) == _lib.NB_ERR_NO_DEVICE and not sh.value
    pos, vel = nb.init_state(8)
    with pytest.raises(nb.NbError):
        nb.NativeShard(pos, vel)
    with pytest.raises(nb.NbError):   # no RCCL without a device either (NB_ERR_NO_DEVICE or NB_ERR_UNSUPPORTED if librccl is absent)
        nb.comm_id()


def test_round5_entry_points_validate_before_touching_the_device(nb):
    """The entry points of round 5 -- the pairs form's phases, the exchanges as pulls, the shard's verify / choose -- check their
    arguments on the host and say NB_ERR_INVALID (or, with sound arguments and no GPU, NB_ERR_NO_DEVICE) before any device work."""
    from nenbody_amd import _lib

    lib = _lib.load()
    fast, strict = nb.default_params(mode=nb.NB_MODE_FAST), nb.default_params()
    # which shapes run their step in phases is host arithmetic: every rank count of config 4, none of config 5, never STRICT
    assert [lib.nb_ring_phased(ctypes.byref(fast), 131072, 0, 131072 // w) for w in (2, 4, 8)] == [1, 1, 1]
    assert [lib.nb_ring_phased(ctypes.byref(fast), 1 << 20, 0, (1 << 20) // w) for w in (2, 4, 8)] == [0, 0, 0]
    assert lib.nb_ring_phased(ctypes.byref(strict), 131072, 0, 16384) == 0 and lib.nb_ring_phased(ctypes.byref(fast), 131072, 100, 16384) == 0
    # the rows' size is the same on every rank of a job (a host may size one buffer for all)
    assert len({lib.nb_ring_scratch_bytes(ctypes.byref(fast), 131072, r * 16384, 16384) for r in range(8)}) == 1
    one = ctypes.c_void_p(16)   # (never dereferenced: the checks come first)
    assert lib.nb_launch_ring_fold_phase(ctypes.byref(fast), 131072, 0, 16384, _lib.NB_RING_OWN, None, one, one, 1 << 30, None) == _lib.NB_ERR_INVALID
    assert lib.nb_launch_ring_fold_phase(ctypes.byref(fast), 131072, 0, 16384, 0, one, one, one, 1 << 30, None) == _lib.NB_ERR_INVALID
    assert lib.nb_launch_ring_fold_phase(ctypes.byref(fast), 131072, 0, 16384, _lib.NB_RING_OWN_READY + 1, one, one, one, 1 << 30, None) == _lib.NB_ERR_INVALID
    assert lib.nb_launch_ring_fold_phase(ctypes.byref(strict), 131072, 0, 16384, _lib.NB_RING_REST, one, one, one, 1 << 30, None) == _lib.NB_ERR_UNSUPPORTED
    assert lib.nb_launch_ring_fold_phase(ctypes.byref(fast), 131072, 0, 16384, _lib.NB_RING_REST, one, one, one, 64, None) == _lib.NB_ERR_INVALID   # scratch too small
    # the fused finish: sums may be NULL (the finish adds the rank's own records), nothing else; never aliased; phased shapes only
    fin = lib.nb_launch_ring_finish_phase
    two = ctypes.c_void_p(32)
    assert fin(ctypes.byref(fast), 131072, 0, 16384, one, one, one, None, one, one, 1 << 30, None) == _lib.NB_ERR_INVALID     # pos_out aliases pos_in
    assert fin(ctypes.byref(fast), 131072, 0, 16384, one, two, one, None, None, one, 1 << 30, None) == _lib.NB_ERR_INVALID    # no recv
    assert fin(ctypes.byref(fast), 131072, 0, 16384, one, two, one, None, one, None, 1 << 30, None) == _lib.NB_ERR_INVALID    # no scratch
    assert fin(ctypes.byref(fast), 131072, 0, 16384, one, two, one, None, one, one, 64, None) == _lib.NB_ERR_INVALID          # scratch too small
    assert fin(ctypes.byref(fast), 1 << 20, 0, 131072, one, two, one, None, one, one, 1 << 40, None) == _lib.NB_ERR_UNSUPPORTED   # config 5: no phases
    assert fin(ctypes.byref(strict), 131072, 0, 16384, one, two, one, None, one, one, 1 << 30, None) == _lib.NB_ERR_UNSUPPORTED
    # the exchanges as pulls
    assert lib.nb_peers_blob_bytes() >= 64 * 5
    p = ctypes.c_void_p()
    for rank, world in ((0, 0), (2, 2), (-1, 2), (0, 17)):
        assert lib.nb_peers_create(rank, world, ctypes.byref(p)) == _lib.NB_ERR_INVALID and not p.value
    assert lib.nb_peers_create(0, 2, None) == _lib.NB_ERR_INVALID
    for fn, args in ((lib.nb_peers_export, (None, None, None, 1, None)), (lib.nb_peers_import, (None, None)), (lib.nb_peers_probe, (None, 10)),
                     (lib.nb_peers_signal, (None, 0, None)), (lib.nb_peers_gather, (None, 0, 0, 16, None)), (lib.nb_peers_ring, (None, 0, 0, None, 16, 1, None)),
                     (lib.nb_shard_peer_export, (None, None)), (lib.nb_shard_peer_import, (None, None)), (lib.nb_shard_use_peers, (None, 1)),
                     (lib.nb_shard_verify_exchanges, (None, None, None)), (lib.nb_shard_pairs_overlapped, (None,))):
        assert fn(*args) == _lib.NB_ERR_INVALID, fn
    assert lib.nb_shard_choose_form(None, 4, None, None) == _lib.NB_ERR_INVALID
    lib.nb_peers_destroy(None)
    assert lib.nb_peers_last_error(None) is not None
    if lib.nb_device_count() == 0:
        assert lib.nb_peers_create(0, 2, ctypes.byref(p)) == _lib.NB_ERR_NO_DEVICE and not p.value
        assert lib.nb_launch_ring_fold_phase(ctypes.byref(fast), 131072, 0, 16384, _lib.NB_RING_OWN, one, one, one, 1 << 30, None) == _lib.NB_ERR_NO_DEVICE


def test_debug_overrides_are_read_once_and_reloaded_on_request(nb, monkeypatch):
    """The NB_* kernel-form overrides are parsed once per process; nb_debug_reload_env() (which the test suite's monkeypatch
    calls for NB_* names) makes the library read them again.  Pure host arithmetic: visible through nb_scratch_bytes."""
    from nenbody_amd import _lib

    lib = _lib.load()
    fast = nb.default_params(mode=nb.NB_MODE_FAST)
    auto = lib.nb_scratch_bytes(ctypes.byref(fast), 131072, 16384)
    os.environ["NB_FAST_SLICES"] = "3"          # behind the library's back: not seen ...
    try:
        assert lib.nb_scratch_bytes(ctypes.byref(fast), 131072, 16384) == auto
        assert lib.nb_debug_reload_env() == _lib.NB_OK  # ... until it is told to look
        assert lib.nb_scratch_bytes(ctypes.byref(fast), 131072, 16384) == (256 + 3 * 4 * 131072 + 64) + 3 * 16384 * 16
    finally:
        del os.environ["NB_FAST_SLICES"]
        lib.nb_debug_reload_env()
    assert lib.nb_scratch_bytes(ctypes.byref(fast), 131072, 16384) == auto
    monkeypatch.setenv("NB_FAST_SLICES", "1")   # the fixture reloads by itself
    monkeypatch.setenv("NB_FAST_GROUPS", "4")
    lib = _lib.load()   # (a legacy form: the fixture has bound libnenbody_hip_legacy.so, which holds it -- tests/conftest.py)
    assert lib.nb_diag_legacy_forms() == 1
    assert lib.nb_scratch_bytes(ctypes.byref(fast), 131072, 16384) == 0   # (an LDS form, named by its groups) the j chunks meet in LDS: nothing through memory
    strict = nb.default_params()
    monkeypatch.setenv("NB_STRICT_BC", "0")
    assert lib.nb_scratch_bytes(ctypes.byref(strict), 131072, 16384) == 0
    monkeypatch.setenv("NB_STRICT_BC", "1")
    assert lib.nb_scratch_bytes(ctypes.byref(strict), 131072, 131072) == 256 + 3 * 4 * 131072 + 64


def test_plan_arithmetic_over_many_shapes(nb):
    """make_plan is pure host arithmetic: walk it over ragged sizes, every mode and tile (this is what the sanitizer build
    of the host code exercises: tests/test_abi_asan.py).  Scratch sizes must be consistent with the shape."""
    from nenbody_amd import _lib

    lib = _lib.load()
    for mode in (nb.NB_MODE_STRICT, nb.NB_MODE_FAST):
        for tile in (0, 256, 512, 1024):
            p = nb.default_params(mode=mode, tile=tile)
            for n in (1, 2, 63, 64, 65, 255, 4095, 4096, 4097, 65535, 65536, 65537, 131072, 1 << 20, (1 << 24) + 5, 1 << 31, 0xffffffff):
                for count in sorted({1, min(n, 64), min(n, 16384), max(1, n // 8), max(1, n // 2), n}):
                    b = lib.nb_scratch_bytes(ctypes.byref(p), n, count)
                    if n > 1 << 31:      # refused (32-bit record indices need headroom for padding): no plan, no scratch
                        assert b == 0
                    elif mode == nb.NB_MODE_FAST:
                        planes = 256 + 3 * 4 * ((n + 63) // 64 * 64) + 64     # the scalar-load form (sets of 4 096 bodies and more, library's own tile)
                        rows = b - planes if (tile == 0 and n >= 4096) else b
                        if tile == 0 and count == n and n % 256 == 0 and 32768 <= n <= 4194304:   # the pairs form
                            w = 8 if n >= 131072 else 4
                            if n <= 262144:      # one tile: three planes of rows, one row per superblock
                                assert rows == 3 * (n // (256 * w)) * n * 4
                            else:                # chunks of 131 072: running sums + the rows of an a side and a b side
                                assert rows == 3 * n * 4 + 2 * 3 * 64 * 131072 * 4
                            continue
                        assert rows >= 0 and rows % (count * 16) == 0 and rows // (count * 16) <= 64
                    else:
                        assert b in (0, 256 + 3 * 4 * ((n + 63) // 64 * 64) + 64)
    # the repeated call of a rank: same shape, same answer (per-thread plan cache), and a different shape in between
    fake_a, fake_b, fake_v = 0x1000, 0x2000, 0x3000
    fast = nb.default_params(mode=nb.NB_MODE_FAST)
    for _ in range(3):
        assert lib.nb_launch_step(ctypes.byref(fast), 131072, 0, 16384, fake_a, fake_b, fake_v, None, 0, None) == _lib.NB_ERR_INVALID
        assert "scratch" in _lib.last_error()
        assert lib.nb_launch_step(ctypes.byref(fast), 1 << 20, 131072, 131072, fake_a, fake_b, fake_v, None, 0, None) == _lib.NB_ERR_INVALID
    # constants the STRICT ladder has no guarded range for still plan (they take the IEEE divide)
    odd = nb.default_params()
    for g, bias in ((0.0, 1e-7), (1e-3, 0.0), (float("inf"), 1e-7), (1e-3, float("nan")), (1e38, 1e-38), (1e-38, 1e38), (-1e-3, 1e-7)):
        odd.G, odd.bias = g, bias
        assert lib.nb_scratch_bytes(ctypes.byref(odd), 131072, 16384) == 256 + 3 * 4 * 131072 + 64


def test_phased_scratch_host_arithmetic(nb):
    """nb_scratch_bytes_phased: a FAST step in two phases through the scalar-load kernel keeps the planes area in front of its
    partial rows (round 3); naming a tile asks for the LDS forms, which need the rows only; STRICT has no phases."""
    from nenbody_amd import _lib

    lib = _lib.load()
    fast = nb.default_params(mode=nb.NB_MODE_FAST)
    n, count, j_lo, j_hi = 131072, 16384, 16384, 32768
    planes = 256 + 3 * 4 * n + 64
    b = lib.nb_scratch_bytes_phased(ctypes.byref(fast), n, count, j_lo, j_hi)
    assert b > planes and (b - planes) % (count * 16) == 0 and 2 <= (b - planes) // (count * 16) <= 64
    tiled = nb.default_params(mode=nb.NB_MODE_FAST, tile=256)
    bt = lib.nb_scratch_bytes_phased(ctypes.byref(tiled), n, count, j_lo, j_hi)
    assert 0 < bt and bt % (count * 16) == 0
    assert lib.nb_scratch_bytes_phased(ctypes.byref(nb.default_params()), n, count, j_lo, j_hi) == 0          # STRICT: refused
    assert lib.nb_scratch_bytes_phased(ctypes.byref(fast), n, count, 5, 3) == 0                                # j_lo > j_hi: refused
    # the whole set as one rank: the one-call step takes the pairs form, its phases the scalar-load fold
    bw = lib.nb_scratch_bytes_phased(ctypes.byref(fast), n, n, 0, n)
    assert bw > planes and (bw - planes) % (n * 16) == 0


def test_diagnostic_entry_points_validate_and_refuse_without_a_device(nb):
    from nenbody_amd import _lib

    lib = _lib.load()
    tf = ctypes.c_double()
    assert lib.nb_selftest_valu_rate(0, 0.0, ctypes.byref(tf), None) == _lib.NB_ERR_INVALID
    assert lib.nb_selftest_valu_rate(0, 0.05, None, None) == _lib.NB_ERR_INVALID
    assert lib.nb_selftest_valu_rate(1, 5.0, ctypes.byref(tf), None) == _lib.NB_ERR_INVALID
    assert lib.nb_selftest_valu_rate(5, 0.05, ctypes.byref(tf), None) == _lib.NB_ERR_INVALID
    assert lib.nb_diag_step_clock(None, 131072, 0.1, None, None, None) == _lib.NB_ERR_INVALID
    assert lib.nb_diag_step_clock(None, 131072, 9.0, ctypes.byref(tf), None, None) == _lib.NB_ERR_INVALID
    assert lib.nb_diag_step_clock(None, 4096, 0.1, ctypes.byref(tf), None, None) == _lib.NB_ERR_UNSUPPORTED   # block chain: no stamps
    buf = ctypes.create_string_buffer(8)
    assert lib.nb_diag_plan(None, 131072, 16384, buf, len(buf)) == _lib.NB_ERR_INVALID and "too small" in _lib.last_error()
    assert _lib.planned_kernels(nb.default_params(), 131072, 131072) == ["step_strict_sl_kernel", "planes_kernel"]
    assert _lib.planned_kernels(nb.default_params(tile=1024), 131072, 131072) == ["step_strict_kernel"]
    assert _lib.planned_kernels(nb.default_params(), 131072, 16384) == ["step_strict_bc_kernel", "planes_kernel"]
    assert _lib.planned_kernels(nb.default_params(), 1000, 1000) == ["step_strict_kernel"]          # small sets: j-parallel
    assert _lib.planned_kernels(nb.default_params(mode=nb.NB_MODE_FAST), 131072, 131072) == ["step_fast_pairs_kernel", "planes_kernel", "pairs_diag_kernel",
                                                                                         "pairs_integrate_kernel"]
    assert _lib.planned_kernels(nb.default_params(mode=nb.NB_MODE_FAST), 16384, 16384)[0] == "step_fast_sl_kernel"       # too few superblock pairs
    assert lib.nb_scratch_bytes(ctypes.byref(nb.default_params(mode=nb.NB_MODE_FAST)), 65536, 65536) == (256 + 3 * 4 * 65536 + 64) + 3 * 64 * 65536 * 4  # superblocks of 1 024
    assert _lib.planned_kernels(nb.default_params(mode=nb.NB_MODE_FAST), 1 << 20, 1 << 20)[0] == "step_fast_pairs_kernel"  # in chunks of 131 072
    assert _lib.planned_kernels(nb.default_params(mode=nb.NB_MODE_FAST), 131072 + 64, 131072 + 64)[0] == "step_fast_sl_kernel"  # not whole blocks
    assert _lib.planned_kernels(nb.default_params(mode=nb.NB_MODE_FAST), 131072, 65536)[0] == "step_fast_sl_kernel"      # a shard: the other body of a pair is elsewhere
    # its scratch: planes + flags, then three planes of rows, one row per superblock of 2 048 bodies
    fastp = nb.default_params(mode=nb.NB_MODE_FAST)
    assert lib.nb_scratch_bytes(ctypes.byref(fastp), 131072, 131072) == (256 + 3 * 4 * 131072 + 64) + 3 * 64 * 131072 * 4
    assert _lib.planned_kernels(nb.default_params(mode=nb.NB_MODE_FAST, tile=256), 131072, 131072) == ["step_fast_wave_kernel", "integrate_partials_kernel"]
    bad = ctypes.c_uint64()
    assert lib.nb_selftest_ladder(1 << 23, 1, ctypes.byref(bad), None) == _lib.NB_ERR_INVALID
    assert lib.nb_selftest_rcp_scaling(5, 4, ctypes.byref(bad)) == _lib.NB_ERR_INVALID
    if lib.nb_device_count() > 0:
        pytest.skip("a HIP device is present")
    assert lib.nb_launch_status(None) == _lib.NB_ERR_NO_DEVICE
    assert lib.nb_selftest_valu_rate(0, 0.05, ctypes.byref(tf), None) == _lib.NB_ERR_NO_DEVICE
    assert lib.nb_diag_step_clock(None, 131072, 0.1, ctypes.byref(tf), None, None) == _lib.NB_ERR_NO_DEVICE
    # the boids drop-in pads the shorter snapshot on the host before it needs the device
    pos, vel = nb.init_state(12)
    with pytest.raises(nb.NbError) as ei:
        nb.update_instance_boids(np.zeros((12, 4, 4), np.float32), pos, pos.copy(), vel[:5].copy(), vel[:5].copy())
    assert ei.value.status == _lib.NB_ERR_NO_DEVICE
    with pytest.raises(nb.NbError):
        nb.update_instance_boids(np.zeros((12, 4, 4), np.float32), pos[:5].copy(), pos[:5].copy(), vel, vel.copy())


def test_the_evidence_stamp_is_of_the_device_code_not_of_the_source_text(nb, tmp_path):
    """kernel_code_sha hashes the offload bundles of the benchmarked kernels inside the built library: it is
    a function of the library file alone, stable from call to call, blind to everything outside `.hip_fatbin` (a copy of the
    library with its host code patched hashes the same) and to the boids unit's bundle, and it moves when a byte of a hashed
    bundle does."""
    from nenbody_amd import _lib

    a = _lib.kernel_code_sha()
    assert len(a) == 16 and a == _lib.kernel_code_sha()
    blob = bytearray(open(_lib.LIB_PATH, "rb").read())
    i = blob.find(b"nb_launch_ring_fold: pos_in, sums and scratch must be non-null")     # a host-side string
    assert i > 0
    blob[i:i + 2] = b"NB"
    host_patched = tmp_path / "host.so"
    host_patched.write_bytes(blob)
    assert _lib.kernel_code_sha(str(host_patched)) == a
    blob = bytearray(open(_lib.LIB_PATH, "rb").read())
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts, j = [], blob.find(magic)
    while j >= 0:
        starts.append(j)
        j = blob.find(magic, j + 1)
    assert len(starts) == 3
    for k, (lo, hi) in enumerate(zip(starts, starts[1:] + [len(blob)])):
        patched = bytearray(blob)
        mid = lo + (min(hi, lo + 600000) - lo) // 2
        patched[mid] ^= 0xFF
        f = tmp_path / f"bundle{k}.so"
        f.write_bytes(patched)
        hashed = b"step_strict_sl_kernel" in blob[lo:hi] or b"planes_kernel" in blob[lo:hi]
        assert (_lib.kernel_code_sha(str(f)) != a) == hashed, f"bundle {k}"


def test_committed_hbm_traffic_is_of_the_current_kernel_sources(nb):
    """bench.py reports roofline.traffic from profiles/hbm_traffic.json only when the file's stamp is the hash of the device code
    that is running (tools/pmc_summary.py --json writes it).  A kernel change without a new rocprofv3 --pmc run leaves the file
    stale -- bench.py then reports traffic = null with the reason; this test makes that visible here."""
    import json

    from nenbody_amd import _lib

    t = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    assert set(t) >= {"n", "count", "kernels", "source"} and t["n"] == 131072 and t["count"] == 131072
    if t.get("code_sha") != _lib.kernel_code_sha():
        pytest.skip("profiles/hbm_traffic.json is stale (the device code of the benchmarked kernels changed since the PMC passes): "
                    "re-run tools/profile_bench.sh and tools/collect_profiles.sh")
    # every kernel the two arithmetics launch per step at this shape, as the library plans it
    planned = set(_lib.planned_kernels(nb.default_params(), 131072, 131072)) | set(_lib.planned_kernels(nb.default_params(mode=nb.NB_MODE_FAST), 131072, 131072))
    assert planned >= {"step_strict_sl_kernel", "step_fast_pairs_kernel", "planes_kernel"}
    for k in sorted(planned):
        assert t["kernels"][k]["bytes_per_launch"] == t["kernels"][k]["read"] + t["kernels"][k]["write"] > 0


def test_camera_constant_is_host_arithmetic_and_matches_the_oracle(oracle):
    """nb_camera_constant = OPENGL_TO_WGPU_MATRIX * cgmath::perspective (gfx.rs:12-17, 365-367): no device needed; the C
    restatement in the oracle gives the same bits, an independent numpy restatement the same values, and cgmath's
    assertions come back as NB_ERR_INVALID."""
    import nenbody_amd as nb
    from nenbody_amd import _lib

    for fov, aspect, near, far in ((45.0, 16 / 9, 1.0, 10000.0), (90.0 / (4 / 3), 4 / 3, 1.0, 10000.0), (25.3125, 1.0, 0.5, 77.0),
                                   (179.0, 2.0, 1e-3, 1e6)):
        got = nb.camera_constant(fov, aspect, near, far)
        ref = oracle.camera_constant(fov, aspect, near, far)
        assert (got.view(np.uint32) == ref.view(np.uint32)).all(), (fov, aspect, near, far)
        f = 1.0 / np.tan(np.deg2rad(np.float32(fov), dtype=np.float64) / 2)
        want = np.zeros((4, 4))                      # [k] = column k of correction * proj
        want[0, 0] = f / aspect
        want[1, 1] = f
        want[2, 2] = 0.5 * (far + near) / (near - far) - 0.5
        want[2, 3] = -1.0
        want[3, 2] = 0.5 * 2 * far * near / (near - far)
        if fov < 170:                                # near 180 degrees tan is ill-conditioned in the binary32 half-angle
            assert np.allclose(got, want, rtol=2e-6, atol=0), (got, want)
    cp = np.zeros(16, np.float32)
    lib = nb.load()
    for bad in ((0.0, 1.0, 1.0, 10.0), (180.0, 1.0, 1.0, 10.0), (-5.0, 1.0, 1.0, 10.0), (45.0, 0.0, 1.0, 10.0), (45.0, 1.0, 0.0, 10.0),
                (45.0, 1.0, 1.0, 1.0), (45.0, 1.0, 1.0, -3.0), (float("nan"), 1.0, 1.0, 10.0)):
        assert lib.nb_camera_constant(*bad, cp.ctypes.data) == _lib.NB_ERR_INVALID, bad
        assert "perspective" in _lib.last_error()
        with pytest.raises(ValueError):
            oracle.camera_constant(*bad)
    assert lib.nb_camera_constant(45.0, 1.0, 1.0, 10.0, None) == _lib.NB_ERR_INVALID


def test_camera_constant_known_answer(oracle):
    """A hand-derived vector neither implementation produced (the library and the oracle share one reading of
    cgmath).  fovy = 90 degrees, aspect 1, near 1, far 3: f = 1 / tan(45 degrees) = 1, so cgmath::perspective is
    diag(1, 1, (3+1)/(1-3) = -2, .) with [2][3] = -1 and [3][2] = 2*3*1/(1-3) = -3, and OPENGL_TO_WGPU_MATRIX (gfx.rs:12-17:
    z' = z/2 + w/2) turns column 2 into (0, 0, -2/2 + -1/2, -1) = (0, 0, -1.5, -1) and column 3 into (0, 0, -3/2, 0): every
    entry but f exactly representable; f is tan of the binary32 nearest to pi/4, within an ulp of 1."""
    import nenbody_amd as nb

    for got in (nb.camera_constant(90.0, 1.0, 1.0, 3.0), oracle.camera_constant(90.0, 1.0, 1.0, 3.0)):
        m = np.asarray(got, np.float32).reshape(4, 4)      # m[k] = column k
        assert abs(float(m[0, 0]) - 1.0) <= 1.2e-7 and m[1, 1] == m[0, 0]
        want = np.array([[m[0, 0], 0, 0, 0], [0, m[0, 0], 0, 0], [0, 0, -1.5, -1.0], [0, 0, -1.5, 0]], np.float32)
        assert (m == want).all(), m
    # aspect 2 halves [0][0] only; far = 7, near = 1: [2][2] = (8/-6)/2 - 1/2, [3][2] = (14/-6)/2 in binary32
    m = np.asarray(nb.camera_constant(90.0, 2.0, 1.0, 7.0), np.float32).reshape(4, 4)
    f = m[1, 1]
    assert m[0, 0] == f / np.float32(2)
    assert m[2, 2] == np.float32(0.5) * (np.float32(8) / np.float32(-6)) + np.float32(-0.5) and m[2, 3] == -1
    assert m[3, 2] == np.float32(0.5) * (np.float32(14) / np.float32(-6)) and m[3, 3] == 0


def test_update_instance_random_validates_without_a_device():
    """the argument checks of the third drop-in run before any device is touched: an empty zip is a no-op (main.rs:386-389
    iterates nothing), a null array with a nonzero length is an error, and a real call without a GPU fails loudly"""
    import nenbody_amd as nb
    from nenbody_amd import _lib

    lib = nb.load()
    p = np.zeros((4, 3), np.float32)
    inst = np.zeros((4, 4, 4), np.float32)
    for fn, tail in ((lib.nb_update_instance_random, ()), (lib.nb_update_instance_random_seeded, (1, 2))):
        assert fn(None, 0, p.ctypes.data, 4, p.ctypes.data, 4, *tail) == _lib.NB_OK
        assert fn(p.ctypes.data, 1, None, 4, p.ctypes.data, 4, *tail) == _lib.NB_ERR_INVALID
        assert "null array" in _lib.last_error()
        if lib.nb_device_count() == 0:
            assert fn(inst.ctypes.data, 4, p.ctypes.data, 4, p.ctypes.data, 4, *tail) == _lib.NB_ERR_NO_DEVICE
    lib.nb_update_random_seed(5)   # host-side state only


def test_graft_entry_build_agrees_with_the_binding():
    """__graft_entry__.build() is what the driver runs every round: its closing check must follow the ABI version of the header
    and the binding (it said `== 1` for a while after the version had moved to 2).  Not a rebuild: the check on the built library."""
    import re

    import __graft_entry__ as g
    from nenbody_amd import _lib

    src = open(g.__file__).read()
    assert "nb_abi_version() == _lib.NB_ABI_VERSION" in src and not re.search(r"nb_abi_version\(\) == \d", src)
    header = open(os.path.join(ROOT, "include", "nenbody.h")).read()
    assert int(re.search(r"#define\s+NB_ABI_VERSION\s+(\d+)", header).group(1)) == _lib.NB_ABI_VERSION == _lib.load().nb_abi_version()


def test_graft_entry_build_runs_to_its_end():
    """the driver's build check, run for real (incremental: the library is up to date, the five small diagnostic tools are
    recompiled): it must come back without raising"""
    import __graft_entry__ as g

    g.build()


def test_a_bare_load_of_the_library_ignores_the_nb_environment(tmp_path):
    """the NB_* kernel-form overrides steer the library only after a diagnostic call has asked for them
    (nb_diag_enable_env / nb_debug_reload_env: the test suite's fixtures, tools/).  In a fresh process with NB_FAST_PAIRS=0,
    NB_TILE=256 and NB_STRICT_SL=0 exported, the plan of the headline shapes is the default one until then.  Host arithmetic only."""
    import subprocess
    import sys

    code = """
import ctypes, sys
sys.path.insert(0, %r)
from nenbody_amd import _lib
lib = _lib.load()
fast, strict = _lib.default_params(_lib.NB_MODE_FAST), _lib.default_params()
def plan(p): return ",".join(_lib.planned_kernels(p, 131072, 131072))
print(plan(fast), plan(strict), lib.nb_ring_partners(ctypes.byref(fast), 131072, 0, 16384))
assert lib.nb_diag_enable_env(1) == 0
print(plan(fast), plan(strict), lib.nb_ring_partners(ctypes.byref(fast), 131072, 0, 16384))
assert lib.nb_diag_enable_env(0) == 0
print(plan(fast), plan(strict), lib.nb_ring_partners(ctypes.byref(fast), 131072, 0, 16384))
""" % ROOT
    env = dict(os.environ, NB_FAST_PAIRS="0", NB_TILE="256", NB_STRICT_SL="0", NB_RING="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    bare, enabled, again = r.stdout.strip().splitlines()
    assert bare == again == "step_fast_pairs_kernel,planes_kernel,pairs_diag_kernel,pairs_integrate_kernel step_strict_sl_kernel,planes_kernel 4"
    assert enabled.split()[0].startswith("step_fast_wave_kernel") and enabled.split()[1] == "step_strict_kernel" and enabled.split()[2] == "0"
