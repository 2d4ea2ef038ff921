"""Generates the golden vectors under tests/golden/ from the CPU oracle (oracle/nbody_oracle.c).

The reference (Rust) ships no tests or fixtures for this path and cannot be built or imported in this
environment, so these vectors pin the ORACLE's outputs (seeded initial state, reference constants
src/main.rs:411-413), not outputs of the reference binary: parity of the oracle itself is unpinned.

    python tests/golden/make_golden.py

Layout of nbody_golden.npz (float32, little-endian):
    n16_init_pos/vel, n16_k{1,10}_pos/vel, n16_k10_inst
    n1024_init is regenerated from the seed (not stored); n1024_k{1,10,100,1000}_pos/vel, n1024_k1000_inst
    n16384_k5_sample_idx / _pos / _vel   64 sampled bodies after 5 steps, + float64 checksums of all bodies

Layout of nbody_golden_c2.npz (BASELINE config 2 at the horizons SURVEY.md section 8d names; `--c2`, about 10 minutes
of CPU on 8 cores):
    n16384_k{100,1000}_sample_idx / _pos / _vel   64 sampled bodies, + XOR of all position / velocity bit patterns

Layout of nbody_golden_c3.npz (BASELINE config 3 = the headline size, N = 131 072, carried through the collapse of the cloud
(about step 40) to the north_star's horizon of 1 000 steps; `--c3`, about half an hour of CPU on 7 threads, checkpointed
under /tmp so that a killed run resumes; `--c3 --k N` stops earlier).  The steps run through the oracle's eight-bodies-per-
vector form (nbo_step_range_batched: the same scalar operations per body), and EVERY step is checked on the spot against the
scalar loop nbo_step_range on the 64 sampled bodies plus 64 bodies drawn afresh each step -- bit for bit, or the run stops:
    n131072_steps                      the step numbers 1..K the per-step checksums belong to
    n131072_xor  [K, 2] uint32         XOR of all position / velocity bit patterns after each step
    n131072_sum  [K, 2] uint32         wrapping uint32 sum of the same bit patterns (catches what an XOR pairs away)
    n131072_sample_idx [64]            sampled bodies
    n131072_k{1,40,100,1000}_sample_pos / _vel
    n131072_k{1,40,100,1000}_nonplanar number of bodies with z != 0 or vz != 0 (the reference's init is planar and stays so)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

SEED = 1234


def main():
    out = {}
    oracle.build(force=True)
    p0, v0 = oracle.init_state(16, SEED)
    out["n16_init_pos"], out["n16_init_vel"] = p0, v0
    for k in (1, 10):
        p, v, inst = oracle.run(p0, v0, k, threads=1, want_instances=True)
        out[f"n16_k{k}_pos"], out[f"n16_k{k}_vel"] = p, v
        if k == 10:
            out["n16_k10_inst"] = inst
    p0, v0 = oracle.init_state(1024, SEED)
    for k in (1, 10, 100, 1000):
        p, v, inst = oracle.run(p0, v0, k, want_instances=True)
        out[f"n1024_k{k}_pos"], out[f"n1024_k{k}_vel"] = p, v
        if k == 1000:
            out["n1024_k1000_inst"] = inst
    p0, v0 = oracle.init_state(16384, SEED)
    p, v = oracle.run(p0, v0, 5)
    idx = np.linspace(0, 16383, 64).astype(np.int64)
    out["n16384_k5_sample_idx"] = idx
    out["n16384_k5_sample_pos"], out["n16384_k5_sample_vel"] = p[idx], v[idx]
    out["n16384_k5_checksum"] = np.array([p.astype(np.float64).sum(), v.astype(np.float64).sum(),
                                          np.abs(p.astype(np.float64)).sum(), np.abs(v.astype(np.float64)).sum()])
    # XOR of all bit patterns: any single-bit difference anywhere shows up
    out["n16384_k5_xor"] = np.array([np.bitwise_xor.reduce(p.view(np.uint32).ravel()),
                                     np.bitwise_xor.reduce(v.view(np.uint32).ravel())], dtype=np.uint32)
    path = os.path.join(ROOT, "tests", "golden", "nbody_golden.npz")
    np.savez_compressed(path, seed=np.array([SEED]), **out)
    print("wrote", path, os.path.getsize(path), "bytes")


def main_c2():
    out = {}
    oracle.build()
    p, v = oracle.init_state(16384, SEED)
    idx = np.linspace(0, 16383, 64).astype(np.int64)
    done = 0
    for k in (100, 1000):
        p, v = oracle.run(p, v, k - done)
        done = k
        out[f"n16384_k{k}_sample_idx"] = idx
        out[f"n16384_k{k}_sample_pos"], out[f"n16384_k{k}_sample_vel"] = p[idx], v[idx]
        out[f"n16384_k{k}_xor"] = np.array([np.bitwise_xor.reduce(p.view(np.uint32).ravel()),
                                            np.bitwise_xor.reduce(v.view(np.uint32).ravel())], dtype=np.uint32)
        print("k", k, "done", flush=True)
    path = os.path.join(ROOT, "tests", "golden", "nbody_golden_c2.npz")
    np.savez_compressed(path, seed=np.array([SEED]), **out)
    print("wrote", path, os.path.getsize(path), "bytes")


def bits_checksums(p, v):
    pu, vu = p.view(np.uint32).ravel(), v.view(np.uint32).ravel()
    return (np.array([np.bitwise_xor.reduce(pu), np.bitwise_xor.reduce(vu)], dtype=np.uint32),
            np.array([pu.sum(dtype=np.uint64) & 0xFFFFFFFF, vu.sum(dtype=np.uint64) & 0xFFFFFFFF], dtype=np.uint32))


def main_c3(k_max=1000, threads=7, ckpt="/tmp/nbody_golden_c3_ckpt.npz"):
    import time
    n = 131072
    horizons = [k for k in (1, 40, 100, 1000) if k <= k_max]
    oracle.build()
    idx = np.linspace(0, n - 1, 64).astype(np.int64)
    out = {"n131072_sample_idx": idx}
    xors, sums = [], []
    p, v = oracle.init_state(n, SEED)
    done = 0
    if os.path.exists(ckpt):
        c = np.load(ckpt)
        if int(c["seed"][0]) == SEED and int(c["n"][0]) == n:
            p, v, done = c["p"], c["v"], int(c["done"][0])
            xors, sums = list(c["xors"]), list(c["sums"])
            for key in c.files:
                if key.startswith("n131072_k"):
                    out[key] = c[key]
            print("resumed at step", done, flush=True)
    t0 = time.time()
    rng = np.random.default_rng(SEED)
    while done < k_max:
        p_old, v_old = p, v
        p, v = oracle.run(p_old, v_old, 1, threads=threads, batched=True)
        done += 1
        for b in np.concatenate([idx, rng.integers(0, n, 64)]):   # the scalar loop on the same snapshot, body by body
            ps, vs = oracle.step_range(p_old, v_old[b:b + 1], int(b), 1)
            if ps.tobytes() != p[b:b + 1].tobytes() or vs.tobytes() != v[b:b + 1].tobytes():
                raise SystemExit(f"step {done}: the batched oracle and the scalar loop differ on body {b}")
        x, s = bits_checksums(p, v)
        xors.append(x)
        sums.append(s)
        if done in horizons:
            out[f"n131072_k{done}_sample_pos"], out[f"n131072_k{done}_sample_vel"] = p[idx].copy(), v[idx].copy()
            out[f"n131072_k{done}_nonplanar"] = np.array([int(np.count_nonzero(p[:, 2]) + np.count_nonzero(v[:, 2]))])
        if done % 10 == 0 or done in horizons:
            np.savez(ckpt + ".tmp.npz", seed=np.array([SEED]), n=np.array([n]), p=p, v=v, done=np.array([done]),
                     xors=np.array(xors), sums=np.array(sums), **{k: a for k, a in out.items() if k.startswith("n131072_k")})
            os.replace(ckpt + ".tmp.npz", ckpt)
            print(f"step {done}  {time.time() - t0:.0f} s  |p|max {np.abs(p).max():.4g}  |v|max {np.abs(v).max():.4g}", flush=True)
    out["n131072_steps"] = np.arange(1, k_max + 1, dtype=np.int64)
    out["n131072_xor"] = np.array(xors, dtype=np.uint32)
    out["n131072_sum"] = np.array(sums, dtype=np.uint32)
    path = os.path.join(ROOT, "tests", "golden", "nbody_golden_c3.npz")
    np.savez_compressed(path, seed=np.array([SEED]), **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    args = sys.argv[1:]
    if "--c3" in args:
        main_c3(k_max=int(args[args.index("--k") + 1]) if "--k" in args else 1000,
                threads=int(args[args.index("--threads") + 1]) if "--threads" in args else 7)
    elif "--c2" in args:
        main_c2()
    else:
        main()
