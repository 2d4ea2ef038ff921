"""bench.py --gpus N started bare: it must start its own N ranks (the driver's N = 1 command was
`python3 bench.py --gpus 1 ...`; the same shape at N > 1 used to exit before touching a GPU)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def test_bare_gpus_2_starts_two_ranks_and_propagates_their_exit_code():
    """No GPU here: every rank ends with 'bench.py needs a HIP device', and the parent must report that failure (non-zero)
    instead of the old 'WORLD_SIZE=1' refusal.  The parent itself never imports torch."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU box: the -m gpu rehearsal below runs the real thing")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-secondary"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "needs a HIP device" in r.stderr
    assert "WORLD_SIZE=1" not in r.stderr


@pytest.mark.gpu
def test_bare_gpus_2_rehearsal_prints_one_json_line():
    """Exactly the driver's command shape at N = 2, on the one GPU of the test box: two ranks share the device and exchange
    through gloo (NB_BENCH_BACKEND=gloo), which the line must own up to in `data`.  No scaling number is claimed."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       env=_env(NB_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1
    assert "REHEARSAL" in line["data"]
    assert line["scaling"] == "strong" and line["value"] > 0
    assert "sharding" in line["config"] and "x2" in line["config"]["sharding"]
    # FAST on two ranks: the ordered fold with its one exchange, and the pairs form on shards with its second (point-to-point) one
    assert line["other_mode"]["mode"] == "fast" and line["other_mode"]["roofline"]["kernel"] == "step_fast_sl_kernel"
    pairs = line["fast_pairs_on_shards"]
    assert pairs["partners"] == 1 and pairs["roofline"]["kernel"] == "step_fast_ring_kernel" and pairs["value"] > 0
    assert 0 < pairs["roofline"]["frac"] < 1 and pairs["roofline"]["frac_nominal"] > pairs["roofline"]["frac"]
    over = line["fast_pairs_on_shards_overlapped"]   # the same in phases, both exchanges behind compute
    assert over["partners"] == 1 and over["value"] > 0 and over["kernel_ms"] > 0
    chosen = line["fast_form_chosen_by_timing"]
    assert chosen["chosen"] in ("pairs", "pairs_overlapped", "ordered") and set(chosen["ms_per_step"]) == {"pairs", "pairs_overlapped", "ordered"}
    # the line proves its own parity: after preheat + warm-up + timed steps on TWO ranks every bit of the state is the CPU oracle's at
    # that step (tests/golden/nbody_golden_c3.npz), and it says what the communicator saw
    pc = line["parity_check"]
    assert pc["bits_equal"] is True and pc["k"] == line["preheat"]["steps"] + 1 + 2 and pc["checksums"]["xor"] == pc["checksums"]["golden_xor"]
    comm = line["comm"]
    assert comm["backend"] == "gloo" and comm["world"] == 2 and comm["distinct_devices"] == 1 and len(comm["device_uuids"]) == 2
    # both kinds of exchange were verified on a pattern and timed; on a shared card the pulls over IPC beat gloo's detour through the host
    assert comm["exchange"]["mapped"] is True and comm["exchange"]["chosen"] in ("peers", "collective") and set(comm["exchange"]["ms_per_step"]) == {"peers", "collective"}
    assert comm["exchange_paths"]["verified"] and comm["exchange_paths"]["all_gather"] in ("peers", "in_place")
    assert set(line["unpreheated"]) >= {"strict", "fast"}
    assert line["boids_controller"]["split_form"]["value"] > 0


@pytest.mark.gpu
def test_gpus_4_mode_fast_rehearsal_takes_the_pairs_form_as_the_headline():
    """`bench.py --gpus 4 --mode fast` on the one GPU of the test box: four ranks share the device, both of the
    step's exchanges go through gloo -- the all-gather and the point-to-point one to the TWO ranks in front.  The headline of that
    line is the pairs form on shards; no scaling number is claimed."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--mode", "fast", "--fast-form", "pairs", "--steps", "2", "--warmup", "1",
                        "--preheat-ms", "20", "--no-cpu-baseline", "--no-secondary"],
                       env=_env(NB_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 4 and "REHEARSAL" in line["data"] and line["config"]["mode"] == "fast"
    assert line["roofline"]["kernel"] == "step_fast_ring_kernel" and "step_fast_ring_kernel" in line["roofline"]["kernels_per_step"][0]
    assert "2 ranks in front" in line["config"]["sharding"]
    # a rank's launch evaluates N^2 / (2 x 4) unordered pairs (+ its blocks' inner pairs once more): frac prices those, frac_nominal N^2 / 4
    n = 131072
    assert 0.95 < line["roofline"]["pair_evaluations_per_launch"] / (n * n / 8) < 1.05
    assert 0 < line["roofline"]["frac"] < 1 and line["roofline"]["frac_nominal"] > 1.8 * line["roofline"]["frac"]
    # FAST's own parity check: one step of this form against one STRICT step, all bodies; both exchanges verified on a pattern first
    pc = line["parity_check"]
    assert pc["within_tolerance"] is True and pc["max_abs_dr"] < 1e-4 and pc["k"] == 1
    assert line["comm"]["exchange_paths"]["verified"] and line["comm"]["exchange_paths"]["ring_exchange"] in ("peers", "grouped")
    assert line["comm"]["world"] == 4 and line["comm"]["distinct_devices"] == 1 and line["comm"]["exchange"]["chosen"] in ("peers", "collective")


@pytest.mark.gpu
def test_gpus_4_mode_fast_rehearsal_chooses_its_form_by_timing_and_runs_the_phases():
    """`bench.py --gpus 4 --mode fast` as the driver would start it (--fast-form auto): after both exchanges were verified on a
    pattern, ShardedScene.choose_form times the pairs form with its exchanges in sequence and behind compute and the ordered fold, the
    timed region runs the fastest, and the line says which and what each cost.  Then the same with the overlapped form named: the
    step in phases as the headline.  (Four ranks: the pool's process guard allows six processes on a card, so the world of eight is
    rehearsed as threads -- tests/test_gpu_native_shard.py -- and over gloo on the CPU -- tests/test_dist_gloo.py.)"""
    base = [sys.executable, BENCH, "--gpus", "4", "--mode", "fast", "--steps", "2", "--warmup", "1", "--preheat-ms", "20", "--no-cpu-baseline", "--no-secondary"]
    r = subprocess.run(base, env=_env(NB_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    form = line["comm"]["fast_form"]
    assert form["chosen"] in ("pairs", "pairs_overlapped", "ordered") and set(form["ms_per_step"]) == {"pairs", "pairs_overlapped", "ordered"}
    ring = line["roofline"]["kernel"] == "step_fast_ring_kernel"
    assert ring == (form["chosen"] != "ordered") and line["parity_check"]["within_tolerance"] is True
    r = subprocess.run(base + ["--fast-form", "pairs_overlapped"], env=_env(NB_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert line["roofline"]["kernel"] == "step_fast_ring_kernel" and "behind compute" in line["config"]["sharding"]
    assert "ring_planes_kernel" in line["roofline"]["kernels_per_step"] and line["roofline"]["kernel_ms"] > 0
    assert line["parity_check"]["within_tolerance"] is True and line["comm"]["fast_form"] is None


@pytest.mark.gpu
def test_gpus_5_strict_ragged_rehearsal_proves_its_parity():
    """`bench.py --gpus 5` over gloo on the one card (five ranks + this process: the pool's process guard allows six on a card): a
    world that does not divide 131 072 -- slots of 26 215 bodies, the last rank short --, every rank in the block chain, the in-place
    all-gather verified on a pattern first, and the line's own parity check: after preheat + warm-up + timed steps every bit of the
    state is the CPU oracle's at that step (STRICT does not depend on the world)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "5", "--steps", "3", "--warmup", "1", "--preheat-ms", "30", "--no-cpu-baseline", "--no-secondary"],
                       env=_env(NB_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == 5 and "x5" in line["config"]["sharding"] and line["config"]["mode"] == "strict"
    pc = line["parity_check"]
    assert pc["bits_equal"] is True and pc["k"] == line["preheat"]["steps"] + 1 + 3
    assert line["comm"]["world"] == 5 and line["comm"]["distinct_devices"] == 1 and len(line["comm"]["device_uuids"]) == 5
    assert line["comm"]["exchange_paths"]["verified"] and line["comm"]["exchange_paths"]["all_gather"] in ("peers", "in_place")
    assert line["comm"]["exchange"]["mapped"] is True
    assert line["roofline"]["kernel"] == "step_strict_bc_kernel"


def _preheat_rank(rank, world, port, out_dir):
    import time

    import torch
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    import bench

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def step():   # as a real step: some work of rank-dependent length, then a collective
        time.sleep(0.002 * (1 + 3 * rank))
        t = torch.ones(1)
        dist.all_reduce(t)
        calls.append(float(t.item()))

    count = bench.preheat(step, torch, dist, world, torch.device("cpu"), 60.0)
    dist.barrier()
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump({"count": count, "calls": len(calls), "sums": sorted(set(calls))}, f)
    dist.destroy_process_group()


def test_preheat_runs_the_same_number_of_steps_on_every_rank(tmp_path):
    """bench.py's preheat steps for a wall-clock span, and every step holds a collective: ranks that timed their own steps and
    chose their own counts would leave one of them waiting in an all-gather forever.  Rank 0 decides; here three gloo ranks
    whose steps take 2, 8 and 14 ms."""
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    world = 3
    mp.spawn(_preheat_rank, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(world)]
    assert len({g["count"] for g in got}) == 1 and got[0]["count"] >= 3          # one count, more than the two timed steps
    assert all(g["calls"] == g["count"] for g in got)
    assert all(g["sums"] == [float(world)] for g in got)                          # every collective met all ranks
