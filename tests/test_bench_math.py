"""bench.py's roofline arithmetic, on the CPU: every printed fraction must mean what it says.  The kernel
times fed in are the ones measured in profiles/r03 and profiles/r04."""
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402  (the parent part of bench.py imports neither torch nor the library)

N = 131072


def test_ordered_folds_price_every_ordered_pair():
    r = bench.roofline_fractions("strict", "planar", "step_strict_sl_kernel", N, N, 5.491)
    assert r["pair_evaluations_per_launch"] == float(N) * N
    assert r["frac"] == r["frac_nominal"] == pytest.approx(18 * N * N / 5.491e-3 / 1e12 / 157.3)
    assert r["frac"] == pytest.approx(0.358, abs=2e-3)
    # 18 full-rate operations + one reciprocal per pair: lane operations per second over 256 CUs x 128 lanes x 2.4 GHz
    assert r["frac_executed"] == pytest.approx(19 * N * N / 5.491e-3 / (256 * 128 * 2.4e9), rel=1e-3)   # (157.3 TFLOP/s / 2)
    assert 0.7 < r["frac_executed"] < 0.8
    shard = bench.roofline_fractions("strict", "planar", "step_strict_bc_kernel", N, N // 8, 0.789)
    assert shard["pair_evaluations_per_launch"] == float(N // 8) * N and shard["frac"] == pytest.approx(0.3113, abs=2e-3)


def test_the_pairs_form_is_priced_on_what_it_executes():
    r = bench.roofline_fractions("fast", "planar", "step_fast_pairs_kernel", N, N, 1.669)
    # every unordered pair of two superblocks once, the pairs inside the 64 superblocks of 2 048 bodies as an ordered fold
    assert r["pair_evaluations_per_launch"] == N * N / 2 + 1024 * N
    assert r["frac"] < 1.0 and r["frac"] == pytest.approx(0.598, abs=3e-3)          # round 3 printed 1.18 under this name
    assert r["frac_nominal"] == pytest.approx(1.178, abs=3e-3) and r["frac_nominal"] > 1.0
    assert r["frac_nominal"] / r["frac"] == pytest.approx(N * N / (N * N / 2 + 1024 * N))
    assert r["frac_executed"] == pytest.approx(5.25 * N * N / 1.669e-3 / 7.8643e13, rel=1e-3) and r["frac_executed"] < 1.0
    small = bench.roofline_fractions("fast", "planar", "step_fast_pairs_kernel", 65536, 65536, 0.5)
    assert small["pair_evaluations_per_launch"] == 65536 * 65536 / 2 + 512 * 65536        # superblocks of 1 024 bodies below 131 072


def test_a_rank_of_the_ring_is_priced_on_its_own_evaluations():
    count = N // 8
    r = bench.roofline_fractions("fast", "planar", "step_fast_ring_kernel", N, count, 0.229)
    assert r["pair_evaluations_per_launch"] == count * N / 2 + 256 * count
    assert r["frac"] < 1.0 and r["frac_nominal"] == pytest.approx(18 * count * N / 0.229e-3 / 1e12 / 157.3)
    assert r["frac"] == pytest.approx(r["frac_nominal"] * (count * N / 2 + 256 * count) / (count * N))
    assert r["executed_per_interaction"]["full_rate_ops"] == 5.0


def test_three_d_data_has_its_own_operation_counts():
    assert bench.roofline_fractions("strict", "3d", "step_strict_sl_kernel", N, N, 7.8)["executed_per_interaction"]["full_rate_ops"] == 26
    assert bench.roofline_fractions("fast", "3d", "step_fast_pairs_kernel", N, N, 2.2)["executed_per_interaction"]["full_rate_ops"] == 7.125
    assert bench.roofline_fractions("fast", "planar", "step_fast_sl_kernel", N, N, 2.3)["frac"] == pytest.approx(0.855, abs=5e-3)


def test_a_zero_kernel_time_does_not_divide():
    r = bench.roofline_fractions("strict", "planar", "step_strict_sl_kernel", N, N, 0.0)
    assert r["frac"] == r["frac_nominal"] == r["frac_executed"] == 0.0


def test_committed_traffic_knows_the_shapes_of_a_multi_gpu_job(tmp_path, monkeypatch):
    """profiles/hbm_traffic.json holds the one-GPU shape and, under "shards", one rank's share at 2 / 4 / 8 ranks; a stamp for other
    device code, an unknown shape or a missing kernel give None and the reason, never a wrong number"""
    import json
    import types

    import bench

    (tmp_path / "profiles").mkdir()
    doc = {"n": 131072, "count": 131072, "code_sha": "abc", "source": "profiles/rNN/pmc/",
           "kernels": {"step_strict_sl_kernel": {"bytes_per_launch": 25}, "planes_kernel": {"bytes_per_launch": 4}},
           "shards": {"16384": {"n": 131072, "count": 16384, "source": "profiles/rNN/pmc_shard/c16384/",
                                "kernels": {"step_strict_bc_kernel": {"bytes_per_launch": 7}, "planes_kernel": {"bytes_per_launch": 4},
                                            "step_fast_ring_kernel": {"bytes_per_launch": 24}}}}}
    (tmp_path / "profiles" / "hbm_traffic.json").write_text(json.dumps(doc))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    nb = types.SimpleNamespace(_lib=types.SimpleNamespace(kernel_code_sha=lambda: "abc"))
    assert bench.committed_traffic(nb, ["step_strict_sl_kernel", "planes_kernel"], 131072, 131072)[0] == 29
    got, why = bench.committed_traffic(nb, ["step_strict_bc_kernel", "planes_kernel"], 131072, 16384)
    assert got == 11 and "pmc_shard/c16384" in why
    assert bench.committed_traffic(nb, ["step_fast_ring_kernel", "planes_kernel"], 131072, 16384)[0] == 28
    assert bench.committed_traffic(nb, ["step_strict_bc_kernel", "planes_kernel"], 131072, 32768)[0] is None      # no such shape
    assert bench.committed_traffic(nb, ["ring_reduce_kernel"], 131072, 16384)[0] is None                          # kernel not measured
    stale = types.SimpleNamespace(_lib=types.SimpleNamespace(kernel_code_sha=lambda: "other"))
    assert bench.committed_traffic(stale, ["step_strict_bc_kernel", "planes_kernel"], 131072, 16384)[0] is None   # other device code


def test_the_bench_lines_own_parity_check(oracle):
    """bench.py's `parity_check`: the final STRICT state of the headline set is held to the golden checksums of
    exactly the step it reached.  Here on the CPU: the oracle's own state after step 1 and step 2 of N = 131 072 passes (the AVX2
    batches are bit-identical to the scalar loop: tests/test_oracle.py), one flipped bit in one velocity fails, a step the file does
    not hold or another set says so instead of claiming anything."""
    import numpy as np

    pos, vel = oracle.init_state(N, 1234)
    p, v = oracle.run(pos, vel, 1, batched=True)
    ok = bench.parity_against_golden(N, 1234, 1, p, v)
    assert ok["bits_equal"] is True and ok["k"] == 1 and ok["checksums"]["xor"] == ok["checksums"]["golden_xor"]
    # the helper's checksums are the test suite's (tests/test_gpu_parity.py:_checksums): XOR and wrapping sum of every word
    u = p.view(np.uint32).ravel()
    assert bench.state_checksums(p, v)[0][0] == int(np.bitwise_xor.reduce(u)) and bench.state_checksums(p, v)[1][0] == int(u.sum(dtype=np.uint64) & 0xFFFFFFFF)
    bad = v.copy()
    bad.view(np.uint32)[77, 1] ^= 1
    assert bench.parity_against_golden(N, 1234, 1, p, bad)["bits_equal"] is False
    assert bench.parity_against_golden(N, 1234, 2, p, v)["bits_equal"] is False          # the right bits of the wrong step
    p2, v2 = oracle.run(p, v, 1, batched=True)
    assert bench.parity_against_golden(N, 1234, 2, p2, v2)["bits_equal"] is True
    for k, n, seed in ((0, N, 1234), (1001, N, 1234), (5, 4096, 1234), (5, N, 99)):
        r = bench.parity_against_golden(n, seed, k, p[:n], v[:n])
        assert r["bits_equal"] is None and "why" in r
