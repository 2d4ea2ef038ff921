// nb_kernels.h -- internal interface between the C ABI (nb_api.hip) and the kernels (nb_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nbk {

// Arguments of one step for bodies [first, first+count) of a set of n_total.
struct StepArgs {
    const float4 *pos_in;  // n_total records (x,y,z,0): the start-of-step snapshot (old_positions, main.rs:415)
    float4 *pos_out;       // n_total records; [first, first+count) written
    float4 *vel;           // count records, local index, updated in place
    float4 *partial;       // FAST with a split j range: [slices][count] partial sums
    uint32_t n_total, first, count;
    float dt, G, bias;     // main.rs:411-413
    uint32_t lo_bits, hi_bits;  // STRICT: bit patterns of the magnitude range where the unscaled divide ladder is exact
    uint32_t force_ieee;        // STRICT: 1 = always take the IEEE '/' path (parameters outside the proven range; tests)
    uint32_t force_3d;          // 2 (= kFlagNonPlanar) = never take the planar (z == 0) shortcut (tests, measurements)
    uint32_t j_chunk;           // FAST: records per j chunk (one per 256-lane group per blockIdx.y slice), a multiple of the tile
    uint32_t no_packed;         // STRICT: 1 = planar tiles take the component-packed fold instead of the j-packed one (tests, measurements)
    uint32_t spin_budget;       // STRICT block chain: polls per wait before a wave gives up; 0 = the default (tests set a tiny one)
    uint32_t bc_prio;           // STRICT block chain: 1 = the wave that waits for / holds a turn runs at a raised priority
    // FAST, a step in two phases (nb_launch_step_phase): the fold runs over j_count VIRTUAL records, virtual index v being
    // record  j = v + j_base, plus hole_len if that is >= hole_lo  -- a contiguous range of the set, or the set without one.
    // A whole-set launch has j_base = 0, hole_len = 0, j_count = n_total.
    uint32_t j_count, j_base, hole_lo, hole_len;
    uint32_t partial_row0;      // FAST: first row of a.partial this launch writes
    uint32_t always_partial;    // FAST: 1 = write partial sums even when gridDim.y == 1 (a later launch integrates)
    // Diagnostic (nb_diag_step_clock only; NULL in every product launch): the first wave of each workgroup of the whole-set
    // kernels stores (s_memtime, s_memrealtime) at entry and exit, 4 words per workgroup: the clock the part holds under
    // THIS kernel is d(s_memtime) / d(s_memrealtime) x 100 MHz.  No output depends on it.
    unsigned long long *stamps;
};

// A radius test `x < T` as ONE vector instruction (nb_boids.inc:boids_below): mask = clamp(fma(-x, k, T * k)) with k a power of
// two chosen by the host so that T * k ~ 2^30: exactly 1.0f where x < T, exactly 0.0f elsewhere (NaN included).
struct BoidsMaskK {
    float k1, tk1;  // rule 1:  d2 < rule_1_distance                    (main.rs:474-475)
    float k2, tk2;  // rule 2:  d2 < the binary32 above t2              (main.rs:485-486: sqrt(d2) < rule_2_distance <=> d2 <= t2)
    float k3, tk3;  // rule 3:  e2 < the binary32 above t3              (main.rs:497-498)
};

// Arguments of one boids step (update_instance_boids, main.rs:443-526) for bodies [first, first+count).
struct BoidsArgs {
    const float4 *pos_in;  // n_total records: old_positions (main.rs:459)
    const float4 *vel_in;  // n_total records: old_velocities (main.rs:460)
    float4 *pos_out;       // n_total records; [first, first+count) written
    float4 *vel_out;       // n_total records; [first, first+count) written
    uint32_t n_total, first, count;
    float dt;              // main.rs:450
    float r1;              // rule_1_distance, compared with the squared distance (main.rs:451, 474-475)
    float t2, t3;          // squared-distance thresholds equivalent to sqrt(d2) < rule_2_distance / rule_3_distance
    float s1, s2, s3;      // rule scales, main.rs:454-456
    uint32_t force_flags;  // OR-ed into every tile's flags: 1 = never the masked-FMA form, 2 = never the planar form, 4 = always test rule 3 (tests)
    uint32_t vlim_bits;    // bit pattern of v_lim: velocity components of at most this magnitude cannot fail the rule-3 test (nb_boids.inc, ALL3)
    float4 *partial;       // split form (launch_boids_split): [slices][count][3] records of partial sums
    uint32_t j_chunk;      // split form: records per slice, a multiple of the tile
    float4 *vtot;          // split form: per-1024-record sums of all velocities (boids_prep_kernel), ceil(n_total / 1024) records
    uint32_t *gflags;      // split form: the step's global flags (non-finite record / velocity above the rule-3 bound); NULL: rule 3 stays in the loops
    uint32_t mk_valid;     // 1 = mk holds usable constants for all three radii (else: dummies, and force_flags bit 1 is set)
    BoidsMaskK mk;         // the three radius tests as single instructions (masked form only; the host clears the form where a radius has no such constants)
};
// form: 0 = one lane per body, 1 = producer/consumer (64 bodies x 16 waves), 2 = one lane per body with (x, y) packed,
// 3 / 4 = chain split (two waves per 64 bodies: rule 1 | rules 2-3) plain / packed
hipError_t launch_boids(const BoidsArgs &a, uint32_t tile, uint32_t form, hipStream_t s);
// the split form (round 4): the j range in `slices` slices of a.j_chunk records, one lane per body per slice, the slices' sums added
// in slice order by a second launch: the reference's predicates and counts, reassociated sums (a tolerance form for small shards)
hipError_t launch_boids_split(const BoidsArgs &a, uint32_t tile, uint32_t slices, hipStream_t s);

hipError_t launch_strict(const StepArgs &a, uint32_t tile, uint32_t unroll, uint32_t lanes, hipStream_t s);
hipError_t launch_strict_jp(const StepArgs &a, uint32_t tile, uint32_t unroll, uint32_t lanes, hipStream_t s);  // nb_kernels.hip, -DNBK_NOSLP_TU
hipError_t launch_strict_pc(const StepArgs &a, uint32_t producers, hipStream_t s);  // producer/consumer form: 64 bodies x (2 + producers) waves per workgroup
// block-chain form for small shards: quotients stay in registers, running sums pass from wave to wave; needs scratch
// `status`: a device word the kernel ORs 1 into when a wave gave up waiting (outputs poisoned): the host's sticky error
hipError_t launch_strict_bc(const StepArgs &a, void *scratch, uint32_t *status, hipStream_t s);
size_t strict_bc_scratch_bytes(uint32_t n_total);
// scalar-load form (nb_nbody_sl.inc): one wave per 64 bodies reads x / y / z planes with scalar loads; same scratch as the block chain
hipError_t launch_strict_sl(const StepArgs &a, uint32_t shape, void *scratch, hipStream_t s);
// the kernel alone (nb_kernels.hip -DNBK_SL_TU): flags / planes as planes_kernel left them; shape: see nb_nbody_sl.inc
hipError_t launch_strict_sl_kernel(const StepArgs &a, uint32_t shape, const uint32_t *flags, uint32_t generation, const float *px,
                                   const float *py, const float *pz, hipStream_t s);
hipError_t launch_fast(const StepArgs &a, uint32_t tile, uint32_t ib, uint32_t groups, uint32_t slices, hipStream_t s);
hipError_t launch_fast_wave(const StepArgs &a, uint32_t tile, uint32_t ib, uint32_t waves, uint32_t slices, hipStream_t s);
// FAST through scalar loads (nb_nbody_sl.inc): whole-set j range, eight waves per workgroup x slices; scratch = planes area
// (strict_bc_scratch_bytes) followed by the slices' partial rows
hipError_t launch_fast_sl(const StepArgs &a, uint32_t ib, uint32_t slices, void *scratch, hipStream_t s);
hipError_t launch_fast_sl_phase(const StepArgs &a, uint32_t ib, uint32_t slices, bool range, uint32_t r0, uint32_t r1, void *scratch,
                                hipStream_t s);  // one phase of a step in two phases (nb_launch_step_phase); rows behind the planes area
hipError_t launch_fast_sl_kernel(const StepArgs &a, uint32_t ib, uint32_t slices, const uint32_t *flags, uint32_t generation,
                                 const float *px, const float *py, const float *pz, hipStream_t s);  // the kernel alone (-DNBK_SL_TU)
// FAST pairs form (nb_nbody_sym.inc): every unordered pair once; whole sets; np = packed pairs of bodies per lane (2 or 4: blocks
// of 128 np bodies; n_total a multiple of that), w = waves per workgroup (8, 4, 2 or 1; np = 4: up to 4): superblocks of 128 np w
// bodies; chunk = bodies per chunk of the two-level walk (0 = the default: one tile up to 262 144 bodies, chunks of 131 072
// beyond; a multiple of the superblock); scratch = planes area (strict_bc_scratch_bytes) followed by fast_pairs_scratch_floats() floats
hipError_t launch_fast_pairs(const StepArgs &a, uint32_t w, uint32_t np, uint32_t chunk, void *scratch, hipStream_t s);
hipError_t launch_fast_pairs_kernels(const StepArgs &a, uint32_t w, uint32_t np, uint32_t chunk, const uint32_t *flags, uint32_t generation,
                                     const float *px, const float *py, const float *pz, float *scratch, hipStream_t s);  // the kernels alone (-DNBK_SL_TU)
uint32_t fast_pairs_rows(uint32_t n_total, uint32_t w, uint32_t np);
size_t fast_pairs_scratch_floats(uint32_t n_total, uint32_t w, uint32_t np, uint32_t chunk);
uint32_t fast_pairs_chunk(uint32_t n_total, uint32_t chunk);  // the chunk a launch will use (>= n_total: one tile)
// FAST pairs form on a SHARD (nb_nbody_ring.inc, "half shell"): the rank's blocks evaluate the pairs with the blocks that follow them
// on the ring, their own sums stay, the other bodies' sums go into `sums` (rank-relative records: [0, count) own, [d count,
// (d + 1) count) for rank + d, d = 1..partners) for the step's second exchange; launch_ring_finish adds what the `partners` ranks
// behind sent (recv: partners * count records, ascending distance) and integrates.  np: packed pairs of bodies per lane (2 or 4);
// ga: a-blocks per launch (0: default), wpb: waves per a-block (0: default).  scratch = planes area + ring_scratch_floats() floats.
// Shapes: count divides n_total (>= 2 ranks), first a multiple of count, count a multiple of 128 np.
uint32_t ring_partners(uint32_t n_total, uint32_t first, uint32_t count, uint32_t np);  // 0: the shape cannot take the form
size_t ring_scratch_floats(uint32_t n_total, uint32_t first, uint32_t count, uint32_t np, uint32_t ga, uint32_t wpb, uint32_t c4_own = 0, uint32_t c4_rest = 0,
                           uint32_t cap = 0);
// the step in PHASES (round 5): can this shape run them (one launch covers the rank)?  what: 1 the pairs inside the rank's own slot
// that one round of workgroups holds, 2 every other pair + the sums of the ranks in front, 3 the rank's own sums; c4_*: sub-tiles
// per workgroup of a phase, cap: sub-tiles of a block's own part that phase 1 takes (0: defaults)
bool ring_phased(uint32_t n_total, uint32_t first, uint32_t count, uint32_t np, uint32_t ga, uint32_t wpb, uint32_t *c4_own = nullptr, uint32_t *c4_rest = nullptr,
                 uint32_t *cap = nullptr);  // the three come back resolved (0 in: choose): a launch that names them skips the search
hipError_t launch_fast_ring_phase(const StepArgs &a, uint32_t np, uint32_t ga, uint32_t wpb, uint32_t c4_own, uint32_t c4_rest, uint32_t cap, uint32_t what,
                                  void *scratch, float4 *sums, hipStream_t s);
hipError_t launch_ring_phase_kernels(const StepArgs &a, uint32_t np, uint32_t ga, uint32_t wpb, uint32_t c4_own, uint32_t c4_rest, uint32_t cap, uint32_t what,
                                     const uint32_t *flags, const float *px, const float *py, const float *pz, float *scratch, float4 *sums,
                                     hipStream_t s);  // the kernels alone (-DNBK_SL_TU)
hipError_t launch_fast_ring(const StepArgs &a, uint32_t np, uint32_t ga, uint32_t wpb, void *scratch, float4 *sums, hipStream_t s);
hipError_t launch_fast_ring_kernels(const StepArgs &a, uint32_t np, uint32_t ga, uint32_t wpb, const uint32_t *flags, uint32_t generation,
                                    const float *px, const float *py, const float *pz, float *scratch, float4 *sums, hipStream_t s);  // the kernels alone (-DNBK_SL_TU)
hipError_t launch_ring_finish(const StepArgs &a, const float4 *sums, const float4 *recv, uint32_t partners, hipStream_t s);
// the phases' finish, fused: adds the rank's own records itself (what == 3 is not launched) and leaves the planes and flag words of the
// new own slot in `scratch` for the next step's first phase, launched as what == 4 (= 1 without its planes launch)
// (sums != NULL: what == 3 ran, its records are read instead)
hipError_t launch_fast_ring_finish_phase(const StepArgs &a, uint32_t np, uint32_t ga, uint32_t wpb, uint32_t c4_own, uint32_t c4_rest, uint32_t cap, void *scratch,
                                         const float4 *sums, const float4 *recv, hipStream_t s);
hipError_t launch_ring_finish_phase_kernel(const StepArgs &a, uint32_t np, uint32_t ga, uint32_t wpb, uint32_t c4_own, uint32_t c4_rest, uint32_t cap,
                                           uint32_t *flags, float *px, float *py, float *pz, float *scratch, const float4 *sums, const float4 *recv,
                                           uint32_t gen_next, hipStream_t s);  // the kernel alone (-DNBK_SL_TU)
// the fixed-order combine of `rows` partial-sum rows + integrate (what launch_fast runs itself after a split whole-set fold)
hipError_t launch_integrate_partials(const StepArgs &a, uint32_t rows, hipStream_t s);
// device_libm: 0 = the angle, its sine and cosine as the host's libm computes them (nb_libm.h: bit-identical matrices), 1 = the device's own
hipError_t launch_instances(uint32_t count, const float4 *pos, const float4 *vel, float4 *inst, hipStream_t s, uint32_t device_libm = 0);
hipError_t launch_cameras(uint32_t count, const float4 *eyes, const float4 *dirs, const float *up3, const float *cp16, float4 *out,
                          hipStream_t s);
hipError_t launch_random(uint32_t first, uint32_t count, float4 *pos, float4 *vel, uint64_t seed, uint64_t step, hipStream_t s);
hipError_t launch_divide_selftest(uint32_t blocks, uint64_t seed, uint32_t per_thread, int d_lo, int d_hi, int n_lo, int n_hi,
                                  unsigned long long *mismatches, float *first_bad, bool single_correction, hipStream_t s);
// div_ladder against '/' for count_md denominator significands x all 2^23 numerator significands (control: unrefined reciprocal)
hipError_t launch_ladder_exhaustive(uint32_t first_md, uint32_t count_md, bool control, unsigned long long *mismatches, float *first_bad,
                                    hipStream_t s);
// v_rcp_f32(m * 2^k) * 2^k == v_rcp_f32(m) for all 2^23 significands m and k in [k_lo, k_hi]
hipError_t launch_rcp_scaling(int k_lo, int k_hi, unsigned long long *violations, hipStream_t s);
// nb_libm.h evaluated on the device: out[i] = bits of fn(float with bit pattern first + i) (0 sinf, 1 cosf, 2 atanf, 3 atan2f(., x = bits `other`))
hipError_t launch_libm_selftest(int fn, uint32_t first, uint32_t count, uint32_t other, uint32_t *out, hipStream_t s);
// `blocks` workgroups of 256 lanes each issue trips * 64 register-only vector instructions per lane (mix 0: v_fma_f32; 1: the folds' mix)
// mix 3 / 4: v_fma_f32 / v_fmac_f32 with DISTINCT source registers per chain (mix 0 shares two sources among all eight chains).
// stamps (may be NULL): 4 words per workgroup, (s_memtime, s_memrealtime) at entry and exit of its first wave
hipError_t launch_valu_stream(int mix, uint32_t blocks, uint32_t trips, float *sink, unsigned long long *stamps, hipStream_t s);
// sharded boids: the gathered staging buffer [world][pos slot | vel slot] -> position and velocity replicas (world * slot records each)
hipError_t launch_unstage(uint32_t slot, uint32_t world, const float4 *stage, float4 *pos, float4 *vel, hipStream_t s);
// the exchanges verified on a pattern (nb_shard_verify_exchanges): record i = (i + 1000 rank + salt, -i, rank, salt), or NaN; the
// check adds the number of records that differ to *mismatches
hipError_t launch_exchange_pattern(float4 *dst, uint32_t count, float rank, float salt, bool fill_nan, hipStream_t s);
hipError_t launch_exchange_check(const float4 *src, uint32_t count, float rank, float salt, uint32_t *mismatches, hipStream_t s);
// the exchanges as pulls (nb_peers.inc): `copies` copies of `records` 16-byte records each, src[k] (another rank's memory) -> dst[k]
constexpr int kPeerMax = 16;   // ranks of one xGMI domain
struct PullArgs {
    const float4 *src[kPeerMax];
    float4 *dst[kPeerMax];
    uint32_t copies, records;
};
hipError_t launch_peer_pull(const PullArgs &a, hipStream_t s);
hipError_t launch_pack(uint32_t count, const float *xyz, float4 *rec, hipStream_t s);
hipError_t launch_unpack(uint32_t count, const float4 *rec, float *xyz, hipStream_t s);
// both stride-3 arrays -> records, and matrices + both record arrays -> stride-3 (null outputs skipped), one launch each
hipError_t launch_import(uint32_t count, const float *pos_xyz, const float *vel_xyz, float4 *pos_rec, float4 *vel_rec, hipStream_t s);
// done.word != NULL (outputs in mapped host memory): the launch's last workgroup writes done.seq there once all outputs are home
struct ExportDone {
    uint32_t *counter = nullptr;  // a zeroed device word (left zero)
    uint32_t *word = nullptr;     // the device's address of the host word to write
    uint32_t seq = 0;
};
hipError_t launch_export(uint32_t count, const float4 *pos_rec, const float4 *vel_rec, float4 *inst, float *pos_xyz, float *vel_xyz,
                         const uint32_t *status_src, uint32_t *status_dst, hipStream_t s, uint32_t device_libm = 0, ExportDone done = ExportDone{});

}  // namespace nbk
