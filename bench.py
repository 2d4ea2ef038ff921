#!/usr/bin/env python3
"""bench.py -- body-updates/s of nenbody's all-pairs gravity + Euler step on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json): N = 131 072 bodies, reference initial distributions (src/main.rs:738-747, seeded),
reference constants (src/main.rs:411-413); a "step" is one update_instance_nbody over the whole set.  With N
GPUs the set is sharded by index range (strong scaling: total work fixed) and positions are all-gathered
once per step (RCCL).  Inputs are resident in HBM before the timed region.

Prints ONE JSON line on rank 0.  `value` = body-updates/s = N * steps / s of the whole job.
`roofline` prices the pair-fold kernel against the fp32 vector peak (the binding roofline: SURVEY.md
section 8d) at the reference's 18 flop per interaction.  `cpu_baseline` is the CPU oracle
(a C restatement of the reference's Rust; kind "port") timed on this host's cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_VECTOR_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_HBM_GBPS = 8000.0            # same guide: HBM3E ~8 TB/s
BYTES_PER_BODY_STEP = 64          # algorithmic HBM bytes: 16-B position + 16-B velocity record, read and written once


def measured_traffic(kernel, n, count):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes, or None if this run's shape differs."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
        if t["n"] == n and t["count"] == count:
            return t["kernels"][kernel]["bytes_per_launch"]
    except Exception:
        pass
    return None
FLOP_PER_INTERACTION = 18         # src/main.rs:428-430 as written: 3 sub, 3 mul + 2 add, 1 add, 3 mul, 3 div, 3 add


def time_mode(nb, torch, dist, args, mode, rank, world, pos, vel):
    params = nb.default_params(mode=mode)
    sc = nb.ShardedScene(pos, vel, params)
    dev = sc.device

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    if world > 1:
        # untimed: bring the communicator and its channels up even when --warmup 0 (the target buffer is the scratch side)
        sc._all_gather_slots(sc.pos[sc.cur ^ 1])
    for _ in range(args.warmup):
        sc.step()
    # kernel-only timing: events on the stream the kernel is launched on (torch's current stream)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    real_step = sc.backend.step

    def timed_step(*a, **kw):
        e0, e1 = ev[timed_step.i]
        timed_step.i += 1
        e0.record()
        real_step(*a, **kw)
        e1.record()

    timed_step.i = 0
    sc.backend.step = timed_step
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sc.step()
    fence()
    t1 = time.perf_counter()
    sc.backend.step = real_step
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    kern_ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev) if sc.count else 0.0
    if mode == nb.NB_MODE_FAST:
        kernel = "step_fast_kernel"
    else:  # nb_api.hip:make_plan: one lane per body above 65 536 bodies per rank, block chain up to there (tiny sets: producer/consumer)
        kernel = "step_strict_kernel" if sc.count > 65536 else ("step_strict_bc_kernel" if sc.n >= 4096 else "step_strict_pc_kernel")
    return {"elapsed_s": float(elapsed.item()), "kernel_ms": kern_ms, "count": sc.count, "n": sc.n, "kernel": kernel}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=131072, help="bodies (BASELINE: 131072)")
    ap.add_argument("--mode", choices=["strict", "fast"], default="strict",
                    help="arithmetic of the headline number: strict = bit-identical to the reference (default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip timing the other mode")
    ap.add_argument("--aux-timeout", type=float, default=180.0,
                    help="seconds the informational legs (other mode, boids, CPU baseline) may take before the headline is "
                         "printed without them")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import nenbody_amd as nb

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: nenbody_amd has no CPU path")
    # NB_BENCH_BACKEND=gloo is a REHEARSAL mode for a one-GPU box (ranks share devices, positions are gathered
    # through the host): it exercises the multi-rank control flow, its numbers mean nothing.
    backend = os.environ.get("NB_BENCH_BACKEND", "nccl")
    device_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    n = args.n
    pos, vel = nb.init_state(n, 1234)
    primary = nb.NB_MODE_STRICT if args.mode == "strict" else nb.NB_MODE_FAST
    res = time_mode(nb, torch, dist, args, primary, rank, world, pos, vel)

    def summarise(r):
        steps_per_s = args.steps / r["elapsed_s"]
        kernel_s = r["kernel_ms"] * 1e-3
        achieved = FLOP_PER_INTERACTION * r["count"] * r["n"] / kernel_s / 1e12 if kernel_s > 0 else 0.0
        traffic = measured_traffic(r["kernel"], r["n"], r["count"])
        roof = {"bound": "fp32_valu", "achieved": achieved, "peak": PEAK_FP32_VECTOR_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_VECTOR_TFLOPS,
                "bound_note": "fp32 vector ALU (SURVEY.md 8d): an all-pairs fp32 fold is neither HBM- nor MFMA-bound; see 'hbm'",
                "traffic": traffic, "traffic_unit": "bytes/launch (HBM, PMC)",
                "algorithmic_bytes_per_launch": BYTES_PER_BODY_STEP * r["count"],
                "kernel": r["kernel"],
                "flop_per_interaction": FLOP_PER_INTERACTION, "interactions_per_launch": float(r["count"]) * r["n"]}
        if traffic is not None and kernel_s > 0:  # how far from the HBM roofline the same launch is
            gbps = traffic / kernel_s / 1e9
            roof["hbm"] = {"achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbps / PEAK_HBM_GBPS}
        return {
            "body_updates_per_s": n * steps_per_s,
            "interactions_per_s": float(n) * n * steps_per_s,
            "ms_per_step": 1e3 / steps_per_s,
            "kernel_ms": r["kernel_ms"],
            "roofline": roof,
        }

    s = summarise(res)
    line = {
        "metric": f"body-updates/sec (N x steps/s) at N={n}, all-pairs gravity + Euler step",
        "value": s["body_updates_per_s"],
        "unit": "body-updates/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": s["ms_per_step"],
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic" if backend == "nccl" or world == 1 else "synthetic (REHEARSAL backend, not a measurement)",
        "config": {"workload": f"N={n} bodies, fp32, reference init distributions (seed 1234), dt=0.1 G=0.001 bias=1e-7",
                   "mode": args.mode, "sharding": f"index range x{world}, all-gather of positions per step",
                   "tile": "library default"},
        "interactions_per_s": s["interactions_per_s"],
        "roofline": s["roofline"],
    }

    # Everything after this point is informational.  If it wedges (a collective that never completes, say), the
    # headline measured above must still be reported: the watchdog prints it and ends this rank.
    import threading

    def give_up():  # pragma: no cover
        if rank == 0:
            line["aux_error"] = f"informational legs did not finish within {args.aux_timeout} s"
            print(json.dumps(line), flush=True)
        os._exit(0)

    watchdog = threading.Timer(args.aux_timeout, give_up)
    watchdog.daemon = True
    watchdog.start()

    if not args.no_secondary:
        other_mode = nb.NB_MODE_FAST if primary == nb.NB_MODE_STRICT else nb.NB_MODE_STRICT
        try:
            o = summarise(time_mode(nb, torch, dist, args, other_mode, rank, world, pos, vel))
            line["other_mode"] = {"mode": "fast" if primary == nb.NB_MODE_STRICT else "strict",
                                  "value": o["body_updates_per_s"], "ms_per_step": o["ms_per_step"], "roofline": o["roofline"]}
        except Exception as e:  # pragma: no cover
            line["other_mode"] = {"error": repr(e)}

    # the boids controller (update_instance_boids, main.rs:443-526), same set and sharding
    try:
        sc = nb.ShardedScene(pos, vel)
        for _ in range(2):
            sc.step_boids()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        bsteps = max(2, min(args.steps, 10))
        t0 = time.perf_counter()
        for _ in range(bsteps):
            sc.step_boids()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        bt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=sc.device)
        if world > 1:
            dist.all_reduce(bt, op=dist.ReduceOp.MAX)
        line["boids_controller"] = {"metric": "body-updates/s, update_instance_boids (bit-exact)",
                                    "value": n * bsteps / float(bt.item()),
                                    "ms_per_step": 1e3 * float(bt.item()) / bsteps, "steps": bsteps}
    except Exception as e:  # pragma: no cover
        line["boids_controller"] = {"error": repr(e)}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle  # cpu_baseline leg: the oracle is timed here, never used by the product path

        cores = oracle.ncores()
        t0 = time.perf_counter()
        oracle.run(pos, vel, 1, threads=cores)
        dt = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": n / dt, "unit": "body-updates/s", "cores": cores, "kind": "port",
                                "sample": f"1 full step of the same N={n} workload ({n * n:.3e} interactions), all host cores, "
                                          "C restatement of src/main.rs:404-441 (-O2 -ffp-contract=off)",
                                "seconds": dt}
    watchdog.cancel()
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
