/*
 * nenbody_diag.h -- diagnostic entry points of libnenbody_hip.so: self-tests, measurement helpers and the switches the
 * parity tests and tools/ use to pin a kernel form.  NOT part of the drop-in boundary (include/nenbody.h): a host that
 * replaces the reference's update functions binds nothing from this file.
 */
#ifndef NENBODY_DIAG_H
#define NENBODY_DIAG_H

#include "nenbody.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostic: checks on the GPU that STRICT's shared-reciprocal division ladder equals the IEEE binary32 divide on
 * `pairs` random (numerator, denominator) pairs drawn over the whole exponent rectangle the range guard admits for
 * `params` (NULL = defaults), structured mantissas included.  *mismatches receives the number of differing results
 * (expected 0); bad_pair, if non-NULL, receives one offending (n, d).  Returns NB_ERR_UNSUPPORTED when the parameters
 * leave no guarded range (STRICT then always uses the IEEE divide). */
int nb_selftest_divide(const nb_params *params, uint64_t pairs, uint64_t seed, uint64_t *mismatches, float *bad_pair);

/* Diagnostic, and the proof STRICT's division rests on.  With every intermediate normal (what the range guard ensures)
 * the ladder commutes with scaling numerator and denominator by powers of two, so its result depends on the two 24-bit
 * significands only.  nb_selftest_ladder compares it with the IEEE divide for `count` denominator significands starting
 * at `first_significand` (both in [0, 2^23); d = 1.significand) against ALL 2^23 numerator significands, on the GPU;
 * (0, 2^23) is the whole space, 7.0e13 pairs, under a minute on one MI355X.  *mismatches receives the number of differing
 * quotients (expected 0); bad_pair, if non-NULL, one offending (n, d).  nb_selftest_rcp_scaling checks the one step that
 * is not IEEE arithmetic: v_rcp_f32(m * 2^k) * 2^k == v_rcp_f32(m) for every significand m and k in [k_lo, k_hi]
 * (-125 <= k_lo <= k_hi <= 125); *violations receives the count (expected 0). */
int nb_selftest_ladder(uint32_t first_significand, uint32_t count, uint64_t *mismatches, float *bad_pair);
int nb_selftest_rcp_scaling(int k_lo, int k_hi, uint64_t *violations);

/* Diagnostic for integrators: do the model matrices (src/main.rs:437-439) of THIS device equal what THIS host's C library computes
 * -- the reference's f32::atan2 / sin_cos bind to it?  The library restates glibc 2.35's atan2f / sinf / cosf for the device
 * (nenbody_amd/csrc/nb_libm.h); on a host with another libm a few arguments in a billion round differently.  `count` (<= 2^26)
 * seeded velocities -- the step's own range, every binade 2^-40 .. 2^40, any direction, near the axes and diagonals -- through
 * nb_launch_instances and through the host's atan2f / sinf / cosf: *mismatches = matrices whose rotation entries differ in some
 * bit (0 on the hosts the library was built for); bad_velocity (may be NULL) receives one offending (x, y). */
int nb_selftest_matrices(uint32_t count, uint64_t seed, uint64_t *mismatches, float *bad_velocity);
/* The same question function by function and argument by argument: fn = 0 sinf, 1 cosf, 2 atanf, 3 atan2f(y = the argument,
 * x = the binary32 with bit pattern x_bits), over `count` consecutive binary32 bit patterns from `first` (first = 0, count = 0: all
 * 2^32) -- the device's restatement against the host's std::sin / cos / atan / atan2, NaN results equal whatever their payload.
 * *mismatches: how many differ; first_bad (may be NULL): the lowest differing bit pattern.  All 2^32 arguments of one function take
 * about a minute (the host side is the slower one). */
int nb_selftest_libm(int fn, uint32_t first, uint64_t count, uint32_t x_bits, uint64_t *mismatches, uint32_t *first_bad);

/* Diagnostic: what the vector ALU of THIS device issues at the clock it holds under load -- a register-only stream of
 * independent instructions on every SIMD (8 waves each) for about `seconds` (<= 2), in lane-operations per second.
 * mix 0: v_fma_f32 only (x 2 flop = the rate the 157.3 TFLOP/s spec peak assumes at 2.4 GHz); mix 1: fma / add / mul / sub in
 * the proportion of the pair folds (same issue slots, less power: the part clocks it higher); mix 2: that mix as v_pk_*
 * instructions, two lane-operations each (what the folds mostly issue); mix 3 / 4: v_fma_f32 / v_fmac_f32 with source
 * registers of their own per chain (mix 0's eight chains share two sources).  clock_mhz, if non-NULL, receives the shader
 * clock the stream ran at: the median over workgroups of d(s_memtime) / d(s_memrealtime) x 100 MHz, stamped by each
 * workgroup's first wave at entry and exit.  bench.py prints them beside the spec peak its roofline fraction is quoted against. */
int nb_selftest_valu_rate(int mix, double seconds, double *lane_ops_per_s, double *clock_mhz);

/* Diagnostic: the shader clock the part holds under the whole-set step kernel of `params` (STRICT: step_strict_sl_kernel, one
 * lane per body through scalar loads; FAST: step_fast_pairs_kernel, or step_fast_sl_kernel where the plan takes the ordered
 * fold) on n bodies of the reference's initial distribution (seed 1234): back-to-back
 * steps for about `seconds` (<= 5), the last of them with the kernel's diagnostic stamps switched on (StepArgs::stamps: the
 * first wave of every workgroup stores s_memtime and s_memrealtime at entry and exit; no output depends on them).
 * clock_mhz: median over workgroups; wave_cycles (may be NULL): median lifetime of those waves in shader cycles; kernel_ms
 * (may be NULL): that step by HIP events.  NB_ERR_UNSUPPORTED for shapes that take another kernel. */
int nb_diag_step_clock(const nb_params *params, uint32_t n, double seconds, double *clock_mhz, double *wave_cycles,
                       double *kernel_ms);

/* Diagnostic: the NB_* environment variables -- kernel-form overrides the parity tests and tools/ use: NB_TILE, NB_FAST_IB,
 * NB_FAST_GROUPS, NB_FAST_WAVES, NB_FAST_SLICES, NB_FAST_NO_SHARE, NB_FAST_SL, NB_FAST_PAIRS / _W / _NP / _CHUNK, NB_RING / _NP / _GA /
 * _WPB, NB_FORCE_3D, NB_STRICT_LANES / _UNROLL / _PC / _BC / _SL / _NO_PACKED / _FORCE_IEEE, NB_BC_SPIN_BUDGET, NB_BC_PRIO,
 * NB_BOIDS_PC / _TILE / _FORCE, NB_INST_DEVICE_LIBM, NB_SELFTEST_CONTROL, NB_DROPIN_ZERO_COPY, NB_DROPIN_POLL -- are NOT read by a process that
 * merely loads the library: nothing a deployment's environment exports steers which kernels run.  nb_diag_enable_env(1) has
 * them read (now, and again at every nb_debug_reload_env()); nb_diag_enable_env(0) forgets them.  nb_debug_reload_env() alone
 * also switches the reading on (a test or tool that has just changed one calls it).  No launch path reads the environment;
 * contexts and shards keep the launch shape chosen when they were created.  (NB_ROCTX = 1 -- roctx ranges around the step
 * loops and the exchanges, for rocprofv3 --marker-trace -- is a feature for hosts, not a kernel form, and is always read.) */
int nb_diag_enable_env(int on);
int nb_debug_reload_env(void);

/* Test-only: with on != 0, nb_shard_use_rccl builds a communicator of ONE rank whatever the shard's world is, and the
 * all-gather then moves nothing -- the whole RCCL call path of a step on a one-GPU box (tests, tools/step_overhead.py).  A
 * process switch, deliberately not an environment variable: nothing a deployment inherits can arm it. */
int nb_diag_rccl_solo(int on);

/* 1: this library holds every launch shape the diagnostic knobs can name (libnenbody_hip_legacy.so: the same sources with
 * -DNB_LEGACY_FORMS, `make -C nenbody_amd/csrc legacy`); 0: the product build, which holds the shapes the library's plan reaches by
 * itself and answers NB_ERR_UNSUPPORTED where a knob names another (producer/consumer and j-parallel STRICT shapes, the
 * workgroup-tile FAST form, wave counts other than eight, ...). */
int nb_diag_legacy_forms(void);

/* on == 1: rank 0's pulls over xGMI (nb_peers_gather / nb_peers_ring) lose their first record -- what a peer's stores not being visible
 * when its flag says so would look like.  on == 2: the first pull of each kind is whole, every later one copies nothing -- a reader served
 * from a cache it filled earlier (verify_exchanges checks the pulls twice from the same buffers for this).  0: off.  A process switch for
 * the test suite: the fallbacks of verify_exchanges on a one-GPU box. */
int nb_diag_peers_lossy(int on);

/* The kernels one step of this shape launches, dominant one first, comma separated ("step_strict_bc_kernel,planes_kernel"),
 * as the library itself plans the launch (make_plan): what bench.py labels its roofline with. */
int nb_diag_plan(const nb_params *params, uint32_t n_total, uint32_t count, char *out, size_t out_bytes);

#ifdef __cplusplus
}
#endif
#endif /* NENBODY_DIAG_H */
