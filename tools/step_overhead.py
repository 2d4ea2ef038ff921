#!/usr/bin/env python3
"""Host + launch + exchange overhead of one multi-GPU step, measured on ONE GPU (diagnostic tool; needs a GPU).

    python -u tools/step_overhead.py [COUNT [N_TOTAL [STEPS]]]          # default: 16384 of 131072, 200 steps

What an 8-GPU run adds to the kernels is (a) the host work per step, (b) the dispatch gaps between the launches of a step
and (c) the all-gather's own launch; (c)'s wire time needs 8 GPUs, everything else does not.  This tool runs rank 0's
share of an 8-rank job -- bodies [0, COUNT) of N_TOTAL -- for STEPS steps with the exchange issued on a communicator of
ONE rank (RCCL's whole call path, nothing on the wire), and reports

    wall/step   host clock around the loop, one synchronisation at the end
    dev/step    device time of the step's kernels (events around the launches, exchange excluded)
    overhead    wall - dev

for ShardedScene (Python: nb_launch_step + torch.distributed, what bench.py --gpus N runs) and NativeShard (the C loop
nb_shard_step(k) with ncclAllGather on the same stream), STRICT and FAST.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import nenbody_amd as nb  # noqa: E402
nb.reload_env()  # tools/ read the NB_* kernel-form knobs; a host that merely loads the library does not (nb_diag_enable_env)

count = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
n = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
world = n // count

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
pos, vel = nb.init_state(n, 1234)


class SoloScene(nb.ShardedScene):
    """rank 0 of `world`, with the exchange issued on the one-rank group: in place on this rank's own slot"""

    def _all_gather_slots(self, buf, async_op=False):
        mine = buf[: self.slot]
        return self.dist.all_gather_into_tensor(mine, mine, async_op=async_op)

    def _ring_exchange_start(self):
        # the pairs form's second exchange: torch refuses a send to oneself, so the one-rank group moves the D chunks through RCCL's
        # all-to-all instead (the same bytes through the same library; the native path below does send to itself)
        return [self.dist.all_to_all_single(self.recv, self.sums[self.count:], async_op=True)], None


def python_path(mode, exchange=True, overlap=False, ring=False, ring_overlap=False):
    sc = SoloScene(pos, vel, nb.default_params(mode=mode), world=world, rank=0, overlap=overlap, ring=ring, ring_overlap=ring_overlap if ring else None)
    if not exchange:
        sc._all_gather_slots = lambda buf, async_op=False: None
        sc._ring_exchange_start = lambda: ([], None)
    if ring:  # two launch calls per step: device time = events around the whole loop with the exchanges off
        assert sc.partners
        for _ in range(10):
            sc.step()
        sc.sync()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(steps):
            sc.step()
        e1.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / steps
        return wall, e0.elapsed_time(e1) / steps * 1e-3
    for _ in range(10):
        sc.step()
    sc.sync()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    real = sc.backend.step

    def timed(*a, **kw):
        e0, e1 = ev[timed.i]
        timed.i += 1
        e0.record()
        real(*a, **kw)
        e1.record()

    # pass 1: wall clock, nothing else in the loop
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        sc.step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps
    if overlap:  # two phases through step_phase: wall time only
        sc.sync()
        return wall, float("nan")
    # pass 2: device time of the step's kernels
    timed.i = 0
    sc.backend.step = timed
    for _ in range(steps):
        sc.step()
    torch.cuda.synchronize()
    sc.backend.step = real
    dev = sum(a.elapsed_time(b) for a, b in ev) / steps * 1e-3
    sc.sync()
    return wall, dev


def native_path(mode, overlap=False, pairs=False):
    nb.load().nb_diag_rccl_solo(1)   # a communicator of one rank whatever `world` is (include/nenbody_diag.h)
    sh = nb.NativeShard(pos, vel, nb.default_params(mode=mode), rank=0, world=world, comm_id=nb.comm_id(), overlap=overlap, pairs=pairs)
    assert bool(sh.partners) == pairs and sh.pairs_overlapped == (pairs and overlap)
    sh.step(10)
    sh.sync()
    t0 = time.perf_counter()
    sh.step(steps)
    sh.sync()
    wall = (time.perf_counter() - t0) / steps
    sh.close()
    nb.load().nb_diag_rccl_solo(0)
    return wall


print(f"rank 0's share: {count} of {n} bodies (world {world}), {steps} steps; exchange = RCCL all-gather on a one-rank communicator")
for name, mode in (("STRICT", nb.NB_MODE_STRICT), ("FAST", nb.NB_MODE_FAST)):
    w_noex, dev = python_path(mode, exchange=False)   # (ring=False: the ordered fold and its one exchange)
    w_py, dev2 = python_path(mode)
    w_c = native_path(mode)
    print(f"{name:6s} dev/step {dev * 1e6:8.1f} us (with the exchange between steps: {dev2 * 1e6:8.1f})", flush=True)
    print(f"{name:6s} ShardedScene, no exchange : wall/step {w_noex * 1e6:8.1f} us  overhead {(w_noex - dev) * 1e6:7.1f} us", flush=True)
    print(f"{name:6s} ShardedScene + all-gather : wall/step {w_py * 1e6:8.1f} us  overhead {(w_py - dev) * 1e6:7.1f} us", flush=True)
    print(f"{name:6s} NativeShard  + all-gather : wall/step {w_c * 1e6:8.1f} us  overhead {(w_c - dev) * 1e6:7.1f} us", flush=True)
    if mode == nb.NB_MODE_FAST:
        # the overlapped form: own-slot phase, wait for the (asynchronous) exchange, rest phase -- on one GPU there is nothing
        # to hide behind the first phase, so this shows what the phasing itself costs
        w_ov, _ = python_path(mode, overlap=True)
        w_cov = native_path(mode, overlap=True)
        print(f"{name:6s} ShardedScene, overlapped  : wall/step {w_ov * 1e6:8.1f} us  ({(w_ov - w_py) * 1e6:+.1f} us against the plain step)", flush=True)
        print(f"{name:6s} NativeShard,  overlapped  : wall/step {w_cov * 1e6:8.1f} us  ({(w_cov - w_c) * 1e6:+.1f} us against the plain step)", flush=True)
        # the pairs form on shards: fold, second exchange, finish, all-gather
        w_rn, dev_r = python_path(mode, exchange=False, ring=True)
        w_r, _ = python_path(mode, ring=True)
        w_cr = native_path(mode, pairs=True)
        print(f"{name:6s} pairs form on shards: dev/step {dev_r * 1e6:8.1f} us (events around the loop, exchanges off)", flush=True)
        print(f"{name:6s} ShardedScene pairs form, no exchange      : wall/step {w_rn * 1e6:8.1f} us", flush=True)
        print(f"{name:6s} ShardedScene pairs form + both exchanges  : wall/step {w_r * 1e6:8.1f} us  (+{(w_r - w_rn) * 1e6:.1f} us)", flush=True)
        print(f"{name:6s} NativeShard  pairs form + both exchanges  : wall/step {w_cr * 1e6:8.1f} us  (+{(w_cr - w_rn) * 1e6:.1f} us)", flush=True)
        # the same step in PHASES (round 5): pairs inside the rank's own slot while the all-gather lands, the second exchange beside the
        # reduce of the rank's own sums -- on one GPU with nothing on the wire: what the split itself costs
        w_on, dev_o = python_path(mode, exchange=False, ring=True, ring_overlap=True)
        w_o, _ = python_path(mode, ring=True, ring_overlap=True)
        w_co = native_path(mode, overlap=True, pairs=True)
        print(f"{name:6s} pairs form in phases: dev/step {dev_o * 1e6:8.1f} us (events around the loop, exchanges off)  ({(dev_o - dev_r) * 1e6:+.1f} us against the one-launch fold)", flush=True)
        print(f"{name:6s} ShardedScene in phases, no exchange       : wall/step {w_on * 1e6:8.1f} us", flush=True)
        print(f"{name:6s} ShardedScene in phases + both exchanges   : wall/step {w_o * 1e6:8.1f} us  ({(w_o - w_r) * 1e6:+.1f} us against the exchanges in sequence)", flush=True)
        print(f"{name:6s} NativeShard  in phases + both exchanges   : wall/step {w_co * 1e6:8.1f} us  ({(w_co - w_cr) * 1e6:+.1f} us against the exchanges in sequence)", flush=True)
dist.destroy_process_group()
