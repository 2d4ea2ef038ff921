// nb_kernels.hip -- gfx950 (MI355X / CDNA4) kernels for nenbody's all-pairs gravity + Euler step.
//
// Path: update_instance_nbody, reference src/main.rs:404-441.  Per body n, against the
// start-of-step snapshot of all positions (old_positions, main.rs:415):
//     a_n = sum_{i=0..N-1} ((p_i - p_n) * G) / (|p_i - p_n|^2 + bias)      main.rs:425-432 (i == n included)
//     v_n = v_n + a_n * dt                                                  main.rs:434
//     p_n = v_n + p_n                      (new v, no dt)                   main.rs:436
//     M_n = T(p_n) * Rz(atan2(v_n.y, v_n.x))                                main.rs:437-439, 141-143
//
// Mapping to the hardware (one thread per body, 64-lane waves, 256-thread workgroups):
//   - the j loop is tiled through LDS: a workgroup stages TJ 16-byte position records per tile
//     (coalesced global_load_dwordx4 -> ds_write_b128, double buffered, one barrier per tile);
//     every lane then walks the tile IN INDEX ORDER reading the same LDS address (a broadcast:
//     no bank conflict), so each wave-instruction evaluates 64 pairs.
//   - the work is fp32 VALU (no MFMA: pairwise math, not a contraction); HBM traffic is 48 B per
//     body per step against 18*N flop per body, so the kernel is VALU-issue bound.
//   - STRICT: the reference's exact binary32 operation order.  The divides by the same denominator share one
//     v_rcp_f32 + Newton step and take ONE FMA correction step each (div_ladder): the compiler's IEEE divide takes two,
//     and an enumeration of all 2^23 x 2^23 significand pairs shows the second never changes the result here; a range
//     guard falls back to the plain IEEE '/' where the unscaled ladder is not covered by that proof.  The fold is packed
//     across consecutive j's (pair2_strict), not across components.  Result: bit-identical.  Shards too small to fill
//     the chip with one lane per body run the block-chain kernel (nb_nbody_bc.inc).
//   - FAST: FMA chain for r^2, v_rcp_f32 for the divide, G hoisted, IB bodies per thread to amortise
//     the LDS reads, j range optionally split over blockIdx.y with a fixed-order combine.
//
// Everything is compiled with -ffp-contract=off: every fused multiply-add is an explicit __builtin_fmaf, every
// unfused a*b+c stays two roundings.
//
// Source layout.  This file is compiled THREE times (Makefile): as is; with -DNBK_NOSLP_TU -fno-slp-vectorize for the kernels
// that lose from SLP packing; and with -DNBK_SL_TU and the machine scheduler switched off for the scalar-load STRICT kernel,
// whose source order is its issue order (nb_nbody_sl.inc).  The kernels live in per-topic includes:
//   nb_nbody_strict.inc  STRICT arithmetic + step_strict_kernel<TJ,U,S>  (S = 1 here; S > 1 "j-parallel" in the SLP-off unit)
//   nb_nbody_pc.inc      STRICT producer/consumer kernel                  (this unit)
//   nb_nbody_bc.inc      STRICT block-chain kernel for small shards        (this unit)
//   nb_nbody_sl.inc      STRICT scalar-load kernel: whole sets and large shards, no LDS, no barrier  (its own unit)
//   nb_nbody_sym.inc     FAST pairs form: every unordered pair evaluated once, whole sets       (the scalar-load unit)
//   nb_nbody_ring.inc    FAST pairs form on a shard ("half shell"): a second exchange carries the other GPUs' halves  (the scalar-load unit)
//   nb_nbody_fast.inc    FAST kernels + fixed-order combine               (this unit)
//   nb_aux.inc           model matrices, cameras, random walk, self-test  (this unit)
//   nb_boids.inc         boids controller, one-lane and producer/consumer (SLP-off unit)
//   nb_launch.inc        host-side launchers
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <type_traits>

#include "nb_kernels.h"

namespace nbk {

static constexpr int kBlock = 256;   // threads per workgroup = 4 waves, one per SIMD of a CU
static constexpr int kWaves = kBlock / 64;
[[maybe_unused]] static inline uint32_t ceil_div_u(uint32_t a, uint32_t b) { return (a + b - 1u) / b; }

#include "nb_nbody_strict.inc"  // both translation units: the one-lane form here, the j-parallel shapes in the SLP-off unit

#if defined(NBK_SL_TU)
#include "nb_nbody_sl.inc"    // alone in its unit: compiled without the machine scheduler (its source order is its issue order)
#include "nb_nbody_sym.inc"   // FAST, every unordered pair once: hand-ordered like the scalar-load folds
#include "nb_nbody_ring.inc"  // the same on a shard of a multi-GPU job: the blocks on a ring, each against the half that follows it
#elif defined(NBK_NOSLP_TU)
#include "nb_boids.inc"
#else
#include "nb_nbody_pc.inc"
#include "nb_nbody_bc.inc"
#include "nb_nbody_fast.inc"
#include "nb_aux.inc"
#endif

#ifndef NBK_SL_TU
#include "nb_launch.inc"
#endif

}  // namespace nbk
