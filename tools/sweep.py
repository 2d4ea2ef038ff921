#!/usr/bin/env python3
"""Throughput sweep of the step kernels over N and launch shape (diagnostic tool; needs a GPU).

    python tools/sweep.py strict            # STRICT over N and tile
    python tools/sweep.py fast              # FAST over N, bodies/thread, j slices, tile
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nenbody_amd as nb  # noqa: E402
nb._lib.use_library(nb._lib.LEGACY_LIB_PATH)   # this script names launch shapes only the legacy build holds (make -C nenbody_amd/csrc legacy)
nb.reload_env()  # tools/ read the NB_* kernel-form knobs; a host that merely loads the library does not (nb_diag_enable_env)


def run(n, mode, steps, env):
    for k, v in env.items():
        os.environ[k] = str(v)
        nb.reload_env()
    pos, vel = nb.init_state(n, 1234)
    with nb.Scene(pos, vel, nb.default_params(mode=mode)) as sc:
        sc.step_n(2)
        sc.sync()
        t0 = time.perf_counter()
        sc.step_n(steps)
        sc.sync()
        dt = (time.perf_counter() - t0) / steps
    for k in env:
        os.environ.pop(k, None)
        nb.reload_env()
    pairs = float(n) * n / dt
    print(f"mode={'strict' if mode == 0 else 'fast'} n={n:8d} {env} ms/step={dt * 1e3:9.3f} pairs/s={pairs:.3e} "
          f"TF18={pairs * 18 / 1e12:6.1f} ({pairs * 18 / 157.3e12 * 100:4.1f}%)", flush=True)


def shard_sweep(n_total=131072):
    """Kernel-only time of one rank's share at world sizes 1..8 (what bounds multi-GPU strong scaling)."""
    import torch

    from nenbody_amd.dist import HipBackend

    be = HipBackend()
    dev = torch.device("cuda", 0)
    pos, vel = nb.init_state(n_total, 1234)
    cur = torch.zeros((n_total, 4)); cur[:, :3] = torch.from_numpy(pos); cur = cur.to(dev)
    nxt = torch.zeros_like(cur)
    for mode in (nb.NB_MODE_STRICT, nb.NB_MODE_FAST):
        for env in ({}, {"NB_FORCE_3D": 1}):
            for k, v in env.items():
                os.environ[k] = str(v)
                nb.reload_env()
            base = None
            for world in (1, 2, 4, 8):
                count = n_total // world
                params = nb.default_params(mode=mode)
                v4 = torch.zeros((count, 4), device=dev)
                sb = be.scratch_bytes(params, n_total, count)
                scratch = torch.empty((sb,), dtype=torch.uint8, device=dev) if sb else None
                for _ in range(2):
                    be.step(params, n_total, 0, count, cur, nxt, v4, scratch)
                torch.cuda.synchronize()
                reps = 5
                t0 = time.perf_counter()
                for _ in range(reps):
                    be.step(params, n_total, 0, count, cur, nxt, v4, scratch)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / reps
                base = base or dt
                print(f"shard mode={'strict' if mode == 0 else 'fast'} {env} world={world} count={count:7d} ms={dt * 1e3:8.3f} "
                      f"compute-only speedup={base / dt:5.2f}", flush=True)
            for k in env:
                os.environ.pop(k, None)
                nb.reload_env()


def lanes_sweep(n_total=131072):
    """STRICT kernel-only time per (shard size, lanes per body, unroll, tile)."""
    import torch

    from nenbody_amd.dist import HipBackend

    be = HipBackend()
    dev = torch.device("cuda", 0)
    pos, vel = nb.init_state(n_total, 1234)
    cur = torch.zeros((n_total, 4)); cur[:, :3] = torch.from_numpy(pos); cur = cur.to(dev)
    nxt = torch.zeros_like(cur)
    for f3d in (0, 1):
        for world in (1, 2, 4, 8):
            count = n_total // world
            for lanes in (1, 2, 4, 8, 16):
                if count * lanes < 65536 * 2 or count * lanes > 131072 * 4:
                    continue
                for unroll in (2, 4, 8):
                    for tile in (256, 1024):
                        if lanes == 1 and unroll == 2 or lanes > 1 and unroll == 8:
                            continue
                        env = {"NB_STRICT_LANES": lanes, "NB_STRICT_UNROLL": unroll, "NB_FORCE_3D": f3d}
                        for k, v in env.items():
                            os.environ[k] = str(v)
                            nb.reload_env()
                        params = nb.default_params(mode=nb.NB_MODE_STRICT, tile=tile)
                        v4 = torch.zeros((count, 4), device=dev)
                        try:
                            for _ in range(2):
                                be.step(params, n_total, 0, count, cur, nxt, v4, None)
                        except Exception as e:
                            continue
                        torch.cuda.synchronize()
                        reps = 4
                        t0 = time.perf_counter()
                        for _ in range(reps):
                            be.step(params, n_total, 0, count, cur, nxt, v4, None)
                        torch.cuda.synchronize()
                        dt = (time.perf_counter() - t0) / reps
                        print(f"3d={f3d} world={world} count={count:7d} lanes={lanes:2d} unroll={unroll} tile={tile:4d} ms={dt * 1e3:8.3f} "
                              f"x{world}={dt * 1e3 * world:7.2f}", flush=True)


def fast_groups_sweep(n_total=131072):
    """FAST: one rank's share at world 1/2/4/8 over (bodies per lane, groups per workgroup, grid.y slices, tile)."""
    import torch

    from nenbody_amd.dist import HipBackend

    be = HipBackend()
    dev = torch.device("cuda", 0)
    pos, vel = nb.init_state(n_total, 1234)
    cur = torch.zeros((n_total, 4)); cur[:, :3] = torch.from_numpy(pos); cur = cur.to(dev)
    nxt = torch.zeros_like(cur)
    # (bodies per lane, groups per workgroup, grid.y slices, tile, waves): waves != 0 = the barrier-free form (groups unused);
    # all None = the library's own choice
    shapes = {131072: [(4, 1, 32, 512, 0), (None,) * 5, (4, 0, 4, 256, 8), (4, 0, 8, 256, 8), (4, 0, 2, 256, 8), (4, 0, 1, 256, 8),
                       (4, 0, 16, 256, 4), (4, 2, 16, 512, 0), (4, 4, 2, 512, 0), (4, 1, 32, 512, 0), (None,) * 5],
              65536: [(4, 1, 32, 512, 0), (None,) * 5, (4, 0, 16, 256, 8)],
              32768: [(4, 1, 64, 512, 0), (None,) * 5, (4, 0, 32, 256, 8)],
              16384: [(2, 1, 64, 512, 0), (None,) * 5, (4, 0, 64, 256, 8), (2, 0, 32, 256, 8)]}
    for count, lst in shapes.items():
        for ib, groups, slices, tile, waves in lst:
            env = {} if ib is None else {"NB_FAST_IB": ib, "NB_FAST_GROUPS": groups, "NB_FAST_SLICES": slices, "NB_TILE": tile,
                                         "NB_FAST_WAVES": waves}
            for k in ("NB_FAST_IB", "NB_FAST_GROUPS", "NB_FAST_SLICES", "NB_TILE", "NB_FAST_WAVES"):
                os.environ.pop(k, None)
            for k, v in env.items():
                os.environ[k] = str(v)
            nb.reload_env()
            params = nb.default_params(mode=nb.NB_MODE_FAST)
            v4 = torch.zeros((count, 4), device=dev)
            sb = be.scratch_bytes(params, n_total, count)
            scratch = torch.zeros((sb,), dtype=torch.uint8, device=dev) if sb else None
            for _ in range(10):
                be.step(params, n_total, 0, count, cur, nxt, v4, scratch)
            torch.cuda.synchronize()
            reps = 20
            t0 = time.perf_counter()
            for _ in range(reps):
                be.step(params, n_total, 0, count, cur, nxt, v4, scratch)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            world = n_total // count
            print(f"fast count={count:7d} of {n_total} ib={ib} groups={groups} waves={waves} slices={slices} tile={tile} scratch={sb / 1e6:6.2f} MB "
                  f"ms={dt * 1e3:8.3f} x{world}={dt * 1e3 * world:7.3f}", flush=True)
    for k in ("NB_FAST_IB", "NB_FAST_GROUPS", "NB_FAST_SLICES", "NB_TILE", "NB_FAST_WAVES"):
        os.environ.pop(k, None)
    nb.reload_env()
    # standalone sets (the whole set in one launch), default shape against bodies per lane and the workgroup-tile form
    for n in (1024, 4096, 8192, 16384, 32768, 1 << 20):
        for env in ({}, {"NB_FAST_IB": 1}, {"NB_FAST_IB": 2}, {"NB_FAST_IB": 4}, {"NB_FAST_WAVES": 0}, {"NB_FAST_WAVES": 4}, {}):
            if n == 1 << 20 and "NB_FAST_IB" in env:
                continue
            run(n, nb.NB_MODE_FAST, max(3, min(200, int(4e11 / (float(n) * n)))), env)


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "strict"
    known = {"strict", "strict2", "fast", "fast2", "fastgroups", "fastsym", "shard", "lanes", "pc", "boids", "boidsshard", "configs", "strictone"}
    if what not in known:
        raise SystemExit(f"unknown sweep {what!r}; one of {sorted(known)}")
    if what == "shard":
        return shard_sweep()
    if what == "fastgroups":
        return fast_groups_sweep()
    if what == "fastsym":
        for n in (16384, 32768, 65536, 131072, 262144):
            # the pairs form (every unordered pair once) against the ordered scalar-load fold
            for env in ({"NB_FAST_PAIRS": 0}, {"NB_FAST_PAIRS": 1}, {"NB_FAST_PAIRS": 0, "NB_FORCE_3D": 1}, {"NB_FAST_PAIRS": 1, "NB_FORCE_3D": 1}):
                run(n, nb.NB_MODE_FAST, max(5, min(200, int(2e12 / (float(n) * n)))), env)
        return
    if what == "lanes":
        return lanes_sweep()
    if what == "strictone":
        # STRICT at the sizes given on the command line (default 131072), shape from the environment
        sizes = [int(x) for x in sys.argv[2:]] or [131072]
        for n in sizes:
            run(n, nb.NB_MODE_STRICT, max(2, min(2000, int(2e11 / (float(n) * n)))), {})
        return
    if what == "boidsshard":
        import torch

        from nenbody_amd.dist import HipBackend

        be = HipBackend()
        dev = torch.device("cuda", 0)
        n_total = 131072
        pos, vel = nb.init_state(n_total, 1234)

        def rec(a):
            t = torch.zeros((n_total, 4))
            t[:, :3] = torch.from_numpy(a)
            return t.to(dev)

        pin, vin = rec(pos), rec(vel)
        pout, vout = torch.zeros_like(pin), torch.zeros_like(vin)
        bp = nb.default_boids_params()
        for f3d in (0, 2):
            for world in (1, 2, 4, 8):
                count = n_total // world
                for pc in (3, 2, 1, 4, 5):  # one lane per body plain / (x, y) packed, producer/consumer, chain split plain / packed
                    os.environ["NB_BOIDS_PC"] = str(pc)
                    nb.reload_env()
                    os.environ["NB_BOIDS_FORCE"] = str(f3d)
                    nb.reload_env()
                    for _ in range(2):
                        be.boids_step(bp, n_total, 0, count, pin, vin, pout, vout)
                    torch.cuda.synchronize()
                    reps = 3
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        be.boids_step(bp, n_total, 0, count, pin, vin, pout, vout)
                    torch.cuda.synchronize()
                    dt = (time.perf_counter() - t0) / reps
                    print(f"boids force={f3d} world={world} count={count:7d} form={pc} ms={dt * 1e3:8.3f} x{world}={dt * 1e3 * world:7.2f}", flush=True)
        for k in ("NB_BOIDS_PC", "NB_BOIDS_FORCE"):
            os.environ.pop(k, None)
        nb.reload_env()
        for count in (131072, 98304, 90112, 81920, 65536, 57344, 53248, 49152, 40960, 32768, 16384):  # the library's own choice
            world = n_total / count
            for _ in range(2):
                be.boids_step(bp, n_total, 0, count, pin, vin, pout, vout)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                be.boids_step(bp, n_total, 0, count, pin, vin, pout, vout)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 5
            print(f"boids default count={count:7d} ms={dt * 1e3:8.3f} x{world:.2f}={dt * 1e3 * world:7.2f}", flush=True)
        return
    if what == "configs":
        # the BASELINE.json configurations on one GPU, library defaults
        for n, steps in ((1024, 1000), (16384, 100), (131072, 20), (1048576, 2)):
            for mode in (nb.NB_MODE_STRICT, nb.NB_MODE_FAST):
                run(n, mode, steps, {})
        return
    if what == "boids":
        for n in (16384, 131072):
            for tile in (256, 512, 1024):
                pos, vel = nb.init_state(n, 1234)
                with nb.Scene(pos, vel) as sc:
                    bp = nb.default_boids_params(tile=tile)
                    sc.step_boids_n(2, bp)
                    sc.sync()
                    steps = 5 if n > 50000 else 20
                    t0 = time.perf_counter()
                    sc.step_boids_n(steps, bp)
                    sc.sync()
                    dt = (time.perf_counter() - t0) / steps
                print(f"boids n={n} tile={tile} ms/step={dt * 1e3:.3f} body-updates/s={n / dt:.3e} pairs/s={float(n) * n / dt:.3e}", flush=True)
        return
    if what == "pc":
        import torch

        from nenbody_amd.dist import HipBackend

        be = HipBackend()
        dev = torch.device("cuda", 0)
        n_total = 131072
        pos, vel = nb.init_state(n_total, 1234)
        cur = torch.zeros((n_total, 4)); cur[:, :3] = torch.from_numpy(pos); cur = cur.to(dev)
        nxt = torch.zeros_like(cur)
        for f3d in (0, 1):
            for world in (1, 2, 4, 8):
                count = n_total // world
                for pc in (0, 8, 14, "bc"):
                    env = {"NB_STRICT_PC": 14 if pc == "bc" else pc, "NB_STRICT_BC": 1 if pc == "bc" else 0, "NB_FORCE_3D": f3d}
                    for k, v in env.items():
                        os.environ[k] = str(v)
                        nb.reload_env()
                    params = nb.default_params(mode=nb.NB_MODE_STRICT)
                    v4 = torch.zeros((count, 4), device=dev)
                    sb = be.scratch_bytes(params, n_total, count)
                    scratch = torch.empty((sb,), dtype=torch.uint8, device=dev) if sb else None
                    for _ in range(2):
                        be.step(params, n_total, 0, count, cur, nxt, v4, scratch)
                    torch.cuda.synchronize()
                    reps = 4
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        be.step(params, n_total, 0, count, cur, nxt, v4, scratch)
                    torch.cuda.synchronize()
                    dt = (time.perf_counter() - t0) / reps
                    print(f"3d={f3d} world={world} count={count:7d} pc={pc} ms={dt * 1e3:8.3f} x{world}={dt * 1e3 * world:7.2f}", flush=True)
        return
    if what == "strict2":
        for env in ({}, {"NB_FORCE_3D": 1}, {"NB_STRICT_UNROLL": 8}, {"NB_FORCE_3D": 1, "NB_STRICT_UNROLL": 8},
                    {"NB_TILE": 1024}, {"NB_TILE": 1024, "NB_STRICT_UNROLL": 8}):
            run(131072, nb.NB_MODE_STRICT, 5, env)
        run(524288, nb.NB_MODE_STRICT, 2, {})
        run(524288, nb.NB_MODE_STRICT, 2, {"NB_FORCE_3D": 1})
    elif what == "fast2":
        for env in ({"NB_FAST_IB": 1, "NB_FAST_SLICES": 16, "NB_TILE": 1024}, {"NB_FAST_IB": 2, "NB_FAST_SLICES": 16, "NB_TILE": 1024},
                    {"NB_FAST_IB": 2, "NB_FAST_SLICES": 16, "NB_TILE": 1024, "NB_FORCE_3D": 1},
                    {"NB_FAST_IB": 2, "NB_FAST_SLICES": 32, "NB_TILE": 512}, {"NB_FAST_IB": 4, "NB_FAST_SLICES": 32, "NB_TILE": 512},
                    {"NB_FAST_IB": 1, "NB_FAST_SLICES": 32, "NB_TILE": 512, "NB_FORCE_3D": 1}, {}):
            run(131072, nb.NB_MODE_FAST, 10, env)
        for env in ({"NB_FAST_IB": 1, "NB_FAST_SLICES": 64, "NB_TILE": 256}, {"NB_FAST_IB": 1, "NB_FAST_SLICES": 32, "NB_TILE": 512},
                    {"NB_FAST_IB": 2, "NB_FAST_SLICES": 64, "NB_TILE": 256}, {}):
            run(16384, nb.NB_MODE_FAST, 20, env)
    elif what == "strict":
        for n in (32768, 65536, 131072, 262144, 524288):
            for tile in (256, 512, 1024):
                run(n, nb.NB_MODE_STRICT, max(2, min(10, int(2e11 / (n * n)))), {"NB_TILE": tile})
        run(131072, nb.NB_MODE_STRICT, 5, {"NB_TILE": 256, "NB_STRICT_FORCE_IEEE": 1})
    else:
        for n in (16384, 131072, 1048576):
            for ib in (1, 2, 4):
                for sl in (1, 2, 4, 8, 16):
                    if n == 1048576 and sl > 2:
                        continue
                    run(n, nb.NB_MODE_FAST, max(2, min(20, int(4e11 / (n * n)))), {"NB_FAST_IB": ib, "NB_FAST_SLICES": sl, "NB_TILE": 512})
        for tile in (256, 1024):
            run(131072, nb.NB_MODE_FAST, 10, {"NB_FAST_IB": 2, "NB_FAST_SLICES": 4, "NB_TILE": tile})


if __name__ == "__main__":
    main()
