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


if __name__ == "__main__":
    main_c2() if "--c2" in sys.argv[1:] else main()
