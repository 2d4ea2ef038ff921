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
//   - STRICT: the reference's exact binary32 operation order.  The three divides by the same
//     denominator share one v_rcp_f32 + Newton step and then run the same FMA correction ladder the
//     compiler's IEEE divide expands to (see div3_shared below); a per-tile range guard falls back to
//     the plain IEEE '/' where the unscaled ladder would not be exact.  Result: bit-identical.
//   - FAST: FMA chain for r^2, v_rcp_f32 for the divide, G hoisted, IB bodies per thread to amortise
//     the LDS reads, j range optionally split over blockIdx.y with a fixed-order combine.
//
// This translation unit is compiled with -ffp-contract=off: every fused multiply-add below is an
// explicit __builtin_fmaf, every unfused a*b+c stays two roundings.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "nb_kernels.h"

namespace nbk {

static constexpr int kBlock = 256;   // threads per workgroup = 4 waves, one per SIMD of a CU
static constexpr int kWaves = kBlock / 64;
[[maybe_unused]] static inline uint32_t ceil_div_u(uint32_t a, uint32_t b) { return (a + b - 1u) / b; }

// ------------------------------------------------------------------------------------------------
// tile staging: global -> registers -> LDS, TJ records per tile, TJ/256 per thread
// ------------------------------------------------------------------------------------------------
template <int TJ>
struct TileRegs {
    float4 r[TJ / kBlock];
};

template <int TJ>
__device__ __forceinline__ void tile_fetch(TileRegs<TJ> &t, const float4 *__restrict__ pos, uint32_t j0, uint32_t jend,
                                           int tid)
{
#pragma unroll
    for (int k = 0; k < TJ / kBlock; ++k) {
        const uint32_t j = j0 + (uint32_t)(k * kBlock + tid);
        t.r[k] = (j < jend) ? pos[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// The LDS record is (x, y, z, bias): the softening constant rides in the unused w slot, so the fold reads
// one ds_read_b128 per j (4 LDS cycles; a 12-byte read costs 8) and needs no separate operand for bias.
template <int TJ>
__device__ __forceinline__ void tile_store(const TileRegs<TJ> &t, float4 *lds_tile, int tid, float bias)
{
#pragma unroll
    for (int k = 0; k < TJ / kBlock; ++k)
        lds_tile[k * kBlock + tid] = make_float4(t.r[k].x, t.r[k].y, t.r[k].z, bias);
}

// ------------------------------------------------------------------------------------------------
// STRICT arithmetic
// ------------------------------------------------------------------------------------------------

// 1 if a coordinate lies outside {0} U [lo, hi] in magnitude (NaN/inf included): then the unscaled
// division ladder is not guaranteed exact and the tile takes the IEEE '/' path.
__device__ __forceinline__ uint32_t coord_oor(float c, uint32_t lo_bits, uint32_t span_bits)
{
    const uint32_t u = __float_as_uint(c) & 0x7fffffffu;
    return (uint32_t)((u != 0u) & ((u - lo_bits) > span_bits));
}

// Three correctly rounded quotients n{x,y,z} / d with ONE reciprocal.
// This is exactly the ladder LLVM's AMDGPU backend emits for an IEEE binary32 divide
//   r0 = rcp(d); e = fma(-d,r0,1); r = fma(e,r0,r0);
//   q0 = n*r; t0 = fma(-d,q0,n); q1 = fma(t0,r,q0); t1 = fma(-d,q1,n); q = fma(t1,r,q1)
// minus v_div_scale / v_div_fmas scaling / v_div_fixup, which are the identity when d, n and n/d are
// normal with headroom -- what coord_oor() + the parameter check in nb_api guarantee.  The reciprocal
// refinement depends on d only, so the three components share it (11 fewer VALU ops per pair).
__device__ __forceinline__ float div_ladder(float n, float d, float r)
{
    const float q0 = n * r;
    const float t0 = __builtin_fmaf(-d, q0, n);
    const float q1 = __builtin_fmaf(t0, r, q0);
    const float t1 = __builtin_fmaf(-d, q1, n);
    return __builtin_fmaf(t1, r, q1);
}

// PLANAR: every z in the tile and every z of this workgroup's bodies is (+/-)0.  Then dz = 0, dz*dz = +0,
// (xx + yy) + 0 == xx + yy, and the z term of the sum is +/-0, which leaves the running sum unchanged:
// skipping the z arithmetic is exact, not an approximation (the reference's own initial state is planar,
// main.rs:740, 745, and stays planar).
template <bool IEEE, bool PLANAR>
__device__ __forceinline__ void pair_strict(const float4 pj, float xi, float yi, float zi, float G, float &qx, float &qy,
                                            float &qz)
{
    // main.rs:428  vec = p_i - p_n   (the reference recomputes the same difference inside distance2)
    const float dx = pj.x - xi, dy = pj.y - yi;
    const float xx = dx * dx, yy = dy * dy;
    float d, dz = 0.f;
    // main.rs:429  dist = ((dx*dx + dy*dy) + dz*dz) + bias          (pj.w holds bias)
    if (PLANAR) {
        d = (xx + yy) + pj.w;
    } else {
        dz = pj.z - zi;
        const float zz = dz * dz;
        d = ((xx + yy) + zz) + pj.w;
    }
    // main.rs:430  (vec * G) / dist, component-wise
    const float nx = dx * G, ny = dy * G;
    if (IEEE) {
        qx = nx / d;
        qy = ny / d;
        qz = PLANAR ? 0.f : (dz * G) / d;
    } else {
        const float r0 = __builtin_amdgcn_rcpf(d);
        const float e = __builtin_fmaf(-d, r0, 1.0f);
        const float r = __builtin_fmaf(e, r0, r0);
        qx = div_ladder(nx, d, r);
        qy = div_ladder(ny, d, r);
        qz = PLANAR ? 0.f : div_ladder(dz * G, d, r);
    }
}

// q of the lane SP places to the right (lane + SP) within a 16-lane DPP row: row_shl:SP.  With all masks enabled and
// bound_ctrl the mov folds into the consuming add (v_add_f32_dpp), so fetching a neighbour's quotient costs no
// instruction of its own and no LDS traffic.
template <int SP>
__device__ __forceinline__ float lane_right(float v)
{
    static_assert(SP >= 1 && SP <= 15, "row_shl:1..15");
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + SP, 0xf, 0xf, true));
}

// sum += q of the body's lane 0, then lane 1, ... lane S-1: the index order of the S j's of a group.  Only the
// FIRST lane of each S-lane group ends up with the body's true running sum (it sees lanes +0..+S-1, all inside its
// group and its DPP row since S <= 16 and groups are S-aligned); the other lanes compute values nobody reads.
template <int S, int SP = 0>
__device__ __forceinline__ void add_in_lane_order(float &sum, float q)
{
    if constexpr (SP == 0) {
        sum = sum + q;
    } else {
        sum = sum + lane_right<SP>(q);
    }
    if constexpr (SP + 1 < S) add_in_lane_order<S, SP + 1>(sum, q);
}

// The fold over one LDS tile, main.rs:425-432.  S = 1: one lane per body walks j in order.  S > 1 ("j-parallel"):
// the S lanes of a body evaluate the S pairs j = g+0 .. g+S-1 of a group concurrently, then the body's first lane
// adds the S quotients in index order (add_in_lane_order), so its running sum sees exactly the reference's sequence
// of additions.  The summation order is untouched; only the independent per-pair work is spread over lanes.
template <bool IEEE, bool PLANAR, int U, int S>
__device__ __forceinline__ void fold_tile_strict(const float4 *tile, int nj, int sl, float xi, float yi, float zi, float G,
                                                 float &sx, float &sy, float &sz)
{
    int g = 0;
    if (S == 1 && !IEEE) {
        // One lane per body, ladder path: two groups of U pairs per trip with the record reads of each group issued a
        // full group ahead (A is read while B is evaluated and vice versa, no register copies), so the ~130-cycle LDS
        // latency is not exposed -- at N = 131 072 only two waves share a SIMD and nothing else would hide it.
        // Same pairs, same order of additions.
        auto eval = [&](const float4 (&rec)[U]) {
            float qx[U], qy[U], qz[U];
#pragma unroll
            for (int u = 0; u < U; ++u) pair_strict<IEEE, PLANAR>(rec[u], xi, yi, zi, G, qx[u], qy[u], qz[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) {  // main.rs:430  sum + term, strictly in index order
                sx = sx + qx[u];
                sy = sy + qy[u];
                if (!PLANAR) sz = sz + qz[u];
            }
        };
        if (2 * U <= nj) {
            float4 ra[U], rb[U];
#pragma unroll
            for (int u = 0; u < U; ++u) ra[u] = tile[u];
            for (; g + 4 * U <= nj; g += 2 * U) {
#pragma unroll
                for (int u = 0; u < U; ++u) rb[u] = tile[g + U + u];
                eval(ra);
#pragma unroll
                for (int u = 0; u < U; ++u) ra[u] = tile[g + 2 * U + u];
                eval(rb);
            }
            // ra holds group g; one more group is guaranteed (g + 2U <= nj)
#pragma unroll
            for (int u = 0; u < U; ++u) rb[u] = tile[g + U + u];
            eval(ra);
            eval(rb);
            g += 2 * U;
        }
    }
    for (; g + U * S <= nj; g += U * S) {
        float qx[U], qy[U], qz[U];
#pragma unroll
        for (int u = 0; u < U; ++u) pair_strict<IEEE, PLANAR>(tile[g + u * S + sl], xi, yi, zi, G, qx[u], qy[u], qz[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) {  // main.rs:430  sum + term, strictly in index order
            add_in_lane_order<S>(sx, qx[u]);
            add_in_lane_order<S>(sy, qy[u]);
            if (!PLANAR) add_in_lane_order<S>(sz, qz[u]);
        }
    }
    for (; g + S <= nj; g += S) {
        float qx, qy, qz;
        pair_strict<IEEE, PLANAR>(tile[g + sl], xi, yi, zi, G, qx, qy, qz);
        add_in_lane_order<S>(sx, qx);
        add_in_lane_order<S>(sy, qy);
        if (!PLANAR) add_in_lane_order<S>(sz, qz);
    }
    if (S > 1 && g < nj) {  // last, partial group of the whole set: lanes past the end contribute +0 (x + 0 == x)
        const bool valid = g + sl < nj;
        float qx, qy, qz;
        pair_strict<IEEE, PLANAR>(tile[valid ? g + sl : g], xi, yi, zi, G, qx, qy, qz);
        qx = valid ? qx : 0.f;
        qy = valid ? qy : 0.f;
        qz = valid ? qz : 0.f;
        add_in_lane_order<S>(sx, qx);
        add_in_lane_order<S>(sy, qy);
        if (!PLANAR) add_in_lane_order<S>(sz, qz);
    }
}

// main.rs:434, 436 -- shared by both modes; unfused on purpose (two roundings each).
__device__ __forceinline__ void integrate(float4 &p, float4 &v, float ax, float ay, float az, float dt)
{
    const float ex = ax * dt, ey = ay * dt, ez = az * dt;
    v.x = v.x + ex;
    v.y = v.y + ey;
    v.z = v.z + ez;
    p.x = v.x + p.x;
    p.y = v.y + p.y;
    p.z = v.z + p.z;
    p.w = 0.f;
    v.w = 0.f;
}

constexpr uint32_t kFlagIeee = 1u;       // a coordinate outside the ladder's proven range -> IEEE '/' for the tile
constexpr uint32_t kFlagNonPlanar = 2u;  // some z != 0

__device__ __forceinline__ uint32_t wave_or(uint32_t v)
{
    return (__any((int)(v & kFlagIeee)) ? kFlagIeee : 0u) | (__any((int)(v & kFlagNonPlanar)) ? kFlagNonPlanar : 0u);
}

__device__ __forceinline__ uint32_t nonzero_bits(float c) { return (__float_as_uint(c) & 0x7fffffffu) != 0u ? kFlagNonPlanar : 0u; }

// S lanes per body (1, 2, 4, 8, 16): a workgroup covers 256/S bodies.  S > 1 multiplies the wave count of a
// shard by S without touching the summation order: that is what lets STRICT fill the chip when a rank owns few bodies.
template <int TJ, int U, int S>
__global__ __launch_bounds__(kBlock) void step_strict_kernel(StepArgs a)
{
    __shared__ float4 tile[2][TJ];
    __shared__ uint32_t tile_flags[2][kWaves];
    __shared__ uint32_t self_flags[kWaves];

    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int sl = tid & (S - 1);                                                       // which of the body's S lanes
    const uint32_t l = blockIdx.x * (uint32_t)(kBlock / S) + (uint32_t)(tid / S);      // body index inside the shard
    const bool live = l < a.count;
    const uint32_t gi = a.first + (live ? l : a.count - 1u);
    const float4 pi = a.pos_in[gi];
    const float xi = pi.x, yi = pi.y, zi = pi.z;

    const uint32_t lo = a.lo_bits, span = a.hi_bits - a.lo_bits;
    {
        uint32_t f = coord_oor(xi, lo, span) | coord_oor(yi, lo, span) | coord_oor(zi, lo, span) | a.force_ieee;
        f |= nonzero_bits(zi) | a.force_3d;
        f = wave_or(f);
        if ((tid & 63) == 0) self_flags[wave] = f;
    }

    TileRegs<TJ> regs;
    const uint32_t n = a.n_total;
    const uint32_t ntiles = (n + (uint32_t)TJ - 1u) / (uint32_t)TJ;
    tile_fetch<TJ>(regs, a.pos_in, 0u, n, tid);
    auto publish = [&](int buf) {
        uint32_t f = 0;
#pragma unroll
        for (int k = 0; k < TJ / kBlock; ++k) {
            f |= coord_oor(regs.r[k].x, lo, span) | coord_oor(regs.r[k].y, lo, span) | coord_oor(regs.r[k].z, lo, span);
            f |= nonzero_bits(regs.r[k].z);
        }
        f = wave_or(f);
        tile_store<TJ>(regs, tile[buf], tid, a.bias);
        if ((tid & 63) == 0) tile_flags[buf][wave] = f;
    };
    publish(0);
    __syncthreads();
    const uint32_t block_self = self_flags[0] | self_flags[1] | self_flags[2] | self_flags[3];

    float sx = 0.f, sy = 0.f, sz = 0.f;  // main.rs:426  Vector3::new(0.0, 0.0, 0.0)
    int buf = 0;
    for (uint32_t t = 0; t < ntiles; ++t) {
        const bool more = (t + 1u) < ntiles;
        if (more) tile_fetch<TJ>(regs, a.pos_in, (t + 1u) * (uint32_t)TJ, n, tid);
        const uint32_t f = block_self | tile_flags[buf][0] | tile_flags[buf][1] | tile_flags[buf][2] | tile_flags[buf][3];
        const uint32_t left = n - t * (uint32_t)TJ;
        const int nj = left < (uint32_t)TJ ? (int)left : TJ;
        if (f == 0u)
            fold_tile_strict<false, true, U, S>(tile[buf], nj, sl, xi, yi, zi, a.G, sx, sy, sz);
        else if ((f & kFlagIeee) == 0u)
            fold_tile_strict<false, false, U, S>(tile[buf], nj, sl, xi, yi, zi, a.G, sx, sy, sz);
        else
            fold_tile_strict<true, false, (S == 1 ? 2 : 1), S>(tile[buf], nj, sl, xi, yi, zi, a.G, sx, sy, sz);
        if (more) publish(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    if (live && sl == 0) {
        float4 p = pi;
        float4 v = a.vel[l];
        integrate(p, v, sx, sy, sz, a.dt);
        a.vel[l] = v;
        a.pos_out[a.first + l] = p;
    }
}

// The nk chunk steps of one tile: step k = work(k, k & 1), then a barrier; `publish` (tile staging) runs before the last
// barrier.  Two steps per loop trip so the ring parity is a literal, and no per-step condition other than the trip
// count: the chunk loop is otherwise bound by the CU's single scalar unit (~35 SALU ops and several branches per wave
// per chunk doubled the step time when roles, paths and tails were decided inside it).
template <class Work, class Publish>
__device__ __forceinline__ void pc_steps(int nk, Work work, Publish publish)
{
    int k = 0;
    for (; k + 2 < nk; k += 2) {  // k stays even; leaves one or two steps
        work(k, 0);
        __syncthreads();
        work(k + 1, 1);
        __syncthreads();
    }
    if (nk - k == 2) {
        work(k, 0);
        __syncthreads();
        ++k;
        work(k, 1);
    } else {
        work(k, 0);
    }
    publish();
    __syncthreads();
}

#ifdef NBK_NOSLP_TU
// ------------------------------------------------------------------------------------------------
// Boids controller: update_instance_boids, main.rs:443-526.
//
// Three index-ordered, predicated folds per body over the snapshots of all positions / all velocities:
//   rule 1 (main.rs:471-480)  centre += p_i, count += 1      if |p_i - p_n|^2      < rule_1_distance && i != n
//   rule 2 (main.rs:482-492)  repel  -= (p_i - p_n)           if sqrt(|p_i - p_n|^2) < rule_2_distance && i != n
//   rule 3 (main.rs:494-504)  match  += v_i, vcount += 1      if sqrt(|v_i - v_n|^2) < rule_3_distance && i != n
// One lane per body, j in index order, so every sum sees the reference's sequence of additions.  The two sqrt
// predicates are evaluated WITHOUT a sqrt: sqrt is monotonic and correctly rounded in the reference, so
// `sqrt(x) < r` holds exactly for x <= T(r), T(r) = the largest binary32 whose correctly rounded root is below r;
// nb_api computes T on the host (bisection over bit patterns with the host's IEEE sqrtf).  NaN fails every test on
// both sides.  The epilogue (means, velocity blend, speed clamp, position update: main.rs:506-521) runs once per
// body with the correctly rounded '/' and sqrt.
//
// Per-tile specialisations, all exact (chosen from flags the loaders compute while staging a tile):
//   SELF    only the tile that contains the workgroup's own bodies tests `i != n`; elsewhere it is true by range.
//   MASKED  every record of the tile finite: a skipped term is added as 0 * x through an FMA with a 0/1 mask --
//           fma(1, x, s) == s + x and fma(0, x, s) == s for finite x (the sums never hold -0: they start at +0 and
//           x + (-x) rounds to +0) -- one compare + one select per rule instead of a select per component.
//           Non-finite records take the select-on-result form, which leaves a skipped sum untouched whatever x is.
//   PLANAR  every z (positions and velocities, tile and own bodies) is +/-0: the z terms contribute +/-0 to the
//           squared distances and leave the z sums at +0, so they are skipped (the reference's initial state is
//           planar, main.rs:740,745, and stays planar).
// This kernel lives in the SLP-off translation unit: packed math buys nothing here and its register shuffles cost 25 %.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kBoidsNonFinite = 1u;  // some coordinate is inf/NaN -> select form
constexpr uint32_t kBoidsNonPlanar = 2u;  // some z != 0 (position or velocity)

__device__ __forceinline__ uint32_t boids_flags(const float4 p, const float4 v)
{
    auto nonfinite = [](float c) { return (__float_as_uint(c) & 0x7f800000u) == 0x7f800000u ? kBoidsNonFinite : 0u; };
    auto nonzero = [](float c) { return (__float_as_uint(c) & 0x7fffffffu) != 0u ? kBoidsNonPlanar : 0u; };
    return nonfinite(p.x) | nonfinite(p.y) | nonfinite(p.z) | nonfinite(v.x) | nonfinite(v.y) | nonfinite(v.z) | nonzero(p.z) |
           nonzero(v.z);
}

struct BoidsAcc {
    float cx, cy, cz, rx, ry, rz, mx, my, mz;  // main.rs:472, 483, 495: all start at Vector3::new(0.0, 0.0, 0.0)
    float cnt, vcnt;                           // counts kept in binary32: exact below 2^24 (nb_api refuses larger sets)
};

template <bool SELF, bool MASKED, bool PLANAR>
__device__ __forceinline__ void boids_pair(BoidsAcc &s, const float4 pj, const float4 vj, uint32_t j, uint32_t gn,
                                           const float4 pn, const float4 vn, float r1, float t2, float t3)
{
    // keep .w "used": the records are then fetched with one ds_read_b128 each (4 LDS cycles) instead of ds_read_b96 (8)
    asm volatile("" ::"v"(pj.w), "v"(vj.w));
    const bool ne = SELF ? (j != gn) : true;                              // main.rs:475 n != i
    const float dx = pj.x - pn.x, dy = pj.y - pn.y;                       // distance2: (other - self)
    const float dz = PLANAR ? 0.f : pj.z - pn.z;
    float d2 = (dx * dx) + (dy * dy);
    if (!PLANAR) d2 = d2 + (dz * dz);
    const bool p1 = (d2 < r1) && ne;                                      // main.rs:474-475 (squared distance)
    const bool p2 = (d2 <= t2) && ne;                                     // main.rs:485-486  sqrt(d2) < rule_2_distance
    const float ex = vj.x - vn.x, ey = vj.y - vn.y;
    const float ez = PLANAR ? 0.f : vj.z - vn.z;
    float e2 = (ex * ex) + (ey * ey);
    if (!PLANAR) e2 = e2 + (ez * ez);
    const bool p3 = (e2 <= t3) && ne;                                     // main.rs:497-498  sqrt(e2) < rule_3_distance
    if (MASKED) {
        const float m1 = p1 ? 1.f : 0.f, m2 = p2 ? -1.f : 0.f, m3 = p3 ? 1.f : 0.f;
        s.cx = __builtin_fmaf(m1, pj.x, s.cx);                            // main.rs:476  sum + p_i
        s.cy = __builtin_fmaf(m1, pj.y, s.cy);
        if (!PLANAR) s.cz = __builtin_fmaf(m1, pj.z, s.cz);
        s.cnt = s.cnt + m1;
        s.rx = __builtin_fmaf(m2, dx, s.rx);                              // main.rs:487  sum - (p_i - p_n)
        s.ry = __builtin_fmaf(m2, dy, s.ry);
        if (!PLANAR) s.rz = __builtin_fmaf(m2, dz, s.rz);
        s.mx = __builtin_fmaf(m3, vj.x, s.mx);                            // main.rs:499  sum + v_i
        s.my = __builtin_fmaf(m3, vj.y, s.my);
        if (!PLANAR) s.mz = __builtin_fmaf(m3, vj.z, s.mz);
        s.vcnt = s.vcnt + m3;
    } else {
        const float c1x = s.cx + pj.x, c1y = s.cy + pj.y, c1z = s.cz + pj.z;
        s.cx = p1 ? c1x : s.cx;
        s.cy = p1 ? c1y : s.cy;
        s.cz = p1 ? c1z : s.cz;
        s.cnt = p1 ? s.cnt + 1.f : s.cnt;
        const float r1x = s.rx - dx, r1y = s.ry - dy, r1z = s.rz - dz;
        s.rx = p2 ? r1x : s.rx;
        s.ry = p2 ? r1y : s.ry;
        s.rz = p2 ? r1z : s.rz;
        const float m1x = s.mx + vj.x, m1y = s.my + vj.y, m1z = s.mz + vj.z;
        s.mx = p3 ? m1x : s.mx;
        s.my = p3 ? m1y : s.my;
        s.mz = p3 ? m1z : s.mz;
        s.vcnt = p3 ? s.vcnt + 1.f : s.vcnt;
    }
}

template <bool SELF, bool MASKED, bool PLANAR>
__device__ __forceinline__ void boids_fold_tile(BoidsAcc &s, const float4 *tp, const float4 *tv, int nj, uint32_t j0, uint32_t gn,
                                                const float4 pn, const float4 vn, float r1, float t2, float t3)
{
    int j = 0;
    for (; j + 16 <= nj; j += 16) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
            boids_pair<SELF, MASKED, PLANAR>(s, tp[j + u], tv[j + u], j0 + (uint32_t)(j + u), gn, pn, vn, r1, t2, t3);
    }
    for (; j < nj; ++j) boids_pair<SELF, MASKED, PLANAR>(s, tp[j], tv[j], j0 + (uint32_t)j, gn, pn, vn, r1, t2, t3);
}

template <int TJ>
__global__ __launch_bounds__(kBlock) void boids_step_kernel(BoidsArgs a)
{
    __shared__ float4 tile_p[2][TJ];
    __shared__ float4 tile_v[2][TJ];
    __shared__ uint32_t tile_flags[2][kWaves];
    __shared__ uint32_t self_flags[kWaves];

    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const uint32_t l = blockIdx.x * (uint32_t)kBlock + (uint32_t)tid;
    const bool live = l < a.count;
    const uint32_t gn = a.first + (live ? l : a.count - 1u);  // this body's global index n
    const float4 pn = a.pos_in[gn];
    const float4 vn = a.vel_in[gn];
    auto wave_flags = [](uint32_t f) {
        return (__any((int)(f & kBoidsNonFinite)) ? kBoidsNonFinite : 0u) | (__any((int)(f & kBoidsNonPlanar)) ? kBoidsNonPlanar : 0u);
    };
    {
        const uint32_t f = wave_flags(boids_flags(pn, vn) | a.force_flags);
        if ((tid & 63) == 0) self_flags[wave] = f;
    }
    // global index range of this workgroup's bodies: only tiles overlapping it can contain i == n
    const uint32_t own_lo = a.first + blockIdx.x * (uint32_t)kBlock;
    const uint32_t own_hi = own_lo + (uint32_t)kBlock;  // exclusive (past-the-shard indices never match a live body)

    TileRegs<TJ> rp, rv;
    const uint32_t n = a.n_total;
    const uint32_t ntiles = (n + (uint32_t)TJ - 1u) / (uint32_t)TJ;
    tile_fetch<TJ>(rp, a.pos_in, 0u, n, tid);
    tile_fetch<TJ>(rv, a.vel_in, 0u, n, tid);
    auto publish = [&](int buf) {
        uint32_t f = 0;
#pragma unroll
        for (int k = 0; k < TJ / kBlock; ++k) {
            f |= boids_flags(rp.r[k], rv.r[k]);
            tile_p[buf][k * kBlock + tid] = rp.r[k];
            tile_v[buf][k * kBlock + tid] = rv.r[k];
        }
        f = wave_flags(f);
        if ((tid & 63) == 0) tile_flags[buf][wave] = f;
    };
    publish(0);
    __syncthreads();
    const uint32_t block_self = self_flags[0] | self_flags[1] | self_flags[2] | self_flags[3];

    BoidsAcc s = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int buf = 0;
    for (uint32_t t = 0; t < ntiles; ++t) {
        const bool more = (t + 1u) < ntiles;
        if (more) {
            tile_fetch<TJ>(rp, a.pos_in, (t + 1u) * (uint32_t)TJ, n, tid);
            tile_fetch<TJ>(rv, a.vel_in, (t + 1u) * (uint32_t)TJ, n, tid);
        }
        const uint32_t j0 = t * (uint32_t)TJ;
        const uint32_t left = n - j0;
        const int nj = left < (uint32_t)TJ ? (int)left : TJ;
        const uint32_t f = block_self | tile_flags[buf][0] | tile_flags[buf][1] | tile_flags[buf][2] | tile_flags[buf][3];
        const bool has_self = j0 < own_hi && j0 + (uint32_t)nj > own_lo;
        const float4 *tp = tile_p[buf], *tv = tile_v[buf];
        if (has_self || (f & kBoidsNonFinite))
            boids_fold_tile<true, false, false>(s, tp, tv, nj, j0, gn, pn, vn, a.r1, a.t2, a.t3);
        else if (f & kBoidsNonPlanar)
            boids_fold_tile<false, true, false>(s, tp, tv, nj, j0, gn, pn, vn, a.r1, a.t2, a.t3);
        else
            boids_fold_tile<false, true, true>(s, tp, tv, nj, j0, gn, pn, vn, a.r1, a.t2, a.t3);
        if (more) publish(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    if (live) {
        float cx = s.cx, cy = s.cy, cz = s.cz, mx = s.mx, my = s.my, mz = s.mz;
        if (s.cnt > 0.f) {  // main.rs:506-508  (count as f32)
            cx = cx / s.cnt;
            cy = cy / s.cnt;
            cz = cz / s.cnt;
        }
        if (s.vcnt > 0.f) {  // main.rs:510-512
            mx = mx / s.vcnt;
            my = my / s.vcnt;
            mz = mz / s.vcnt;
        }
        // main.rs:514  vel = center*s1 + repel*s2 + match*s3, left to right, every product and sum rounded
        const float ax = cx * a.s1, ay = cy * a.s1, az = cz * a.s1;
        const float bx = s.rx * a.s2, by = s.ry * a.s2, bz = s.rz * a.s2;
        const float gx = mx * a.s3, gy = my * a.s3, gz = mz * a.s3;
        float vx = (ax + bx) + gx, vy = (ay + by) + gy, vz = (az + bz) + gz;
        // main.rs:516-518  |vel| > 1 -> vel * (1 / |vel|).  __builtin_sqrtf is correctly rounded under
        // -fhip-fp32-correctly-rounded-divide-sqrt (__fsqrt_rn is NOT: it is the 1-ulp v_sqrt_f32).
        const float q0 = vx * vx, q1 = vy * vy, q2 = vz * vz;
        const float mag = __builtin_sqrtf((q0 + q1) + q2);
        if (mag > 1.0f) {
            const float sc = 1.0f / mag;
            vx = vx * sc;
            vy = vy * sc;
            vz = vz * sc;
        }
        // main.rs:521  pos = vel * dt + pos
        const float sx = vx * a.dt, sy = vy * a.dt, sz = vz * a.dt;
        a.vel_out[a.first + l] = make_float4(vx, vy, vz, 0.f);
        a.pos_out[a.first + l] = make_float4(sx + pn.x, sy + pn.y, sz + pn.z, 0.f);
    }
}

// ------------------------------------------------------------------------------------------------
// Boids, producer/consumer form ("bpc"), for shards too small to fill the chip with one lane per body.  Boids has
// EIGHT serial chains per body (centre x/y, count, repel x/y, match x/y, vcount; three more in 3-D), so the consumer
// side is one wave per chain.  A workgroup owns 64 bodies (lane = body) and runs 16 waves:
//   waves 0..7   CONSUMERS, one running sum each (main.rs:476, 487, 499), folding the previous 64-j chunk in j order:
//                sum = fma(mask_j, src_j, sum) with the 0/1 (or 0/-1) masks the producers left in the LDS ring and
//                src_j read straight from the SoA tile (a broadcast), four j's per ds_read_b128 on both
//   waves 8..15  PRODUCERS, 8 consecutive j's each per chunk: squared distances and the three radius tests
//                (main.rs:474-475, 485-486, 497-498), written as masks m1, m2 (-1/0), m3 to ring[parity][plane][body][j]
// Only masks cross the ring (3 planes, 54 KB), one barrier per chunk, tiles of 8 chunks staged as SoA planes
// (px, py, pz, vx, vy, vz) one tile ahead by the producers.  Same arithmetic as boids_pair: same bits.
// (Chunks are 64 j's, 8 per producer: the masks travel as binary16, see BpcRing.)
// Per-tile forms as in boids_step_kernel: the self test only in the tile holding the workgroup's bodies, select-on-
// result instead of the masked FMA when a record is not finite, z chains only when some z is non-zero.
// ------------------------------------------------------------------------------------------------
constexpr int kBpcConsumers = 8;
constexpr int kBpcProducers = 8;
constexpr int kBpcThreads = (kBpcConsumers + kBpcProducers) * 64;
constexpr int kBpcPer = 8;                          // consecutive j's per producer per chunk
constexpr int kBpcChunk = kBpcProducers * kBpcPer;  // 64
constexpr int kBpcStride = kBpcChunk + 8;           // halfs per body row: 144 B, rows stay 16-B aligned and spread over the banks
constexpr int kBpcTile = 512;
constexpr int kBpcChunksPerTile = kBpcTile / kBpcChunk;  // 8

// The masks are 0, 1 or -1: exact in binary16, and v_fma_mix_f32 takes a binary16 multiplicand straight into an fp32 FMA,
// so half-width masks cost the consumers nothing and let a chunk (one barrier) cover 64 j's in the same 54 KB of LDS.
struct BpcRing {
    _Float16 m[3][64][kBpcStride];  // masks of rule 1, 2 (negated), 3: [plane][body][j within chunk]
};
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
struct BpcTile {
    float c[6][kBpcTile];  // px, py, pz, vx, vy, vz
};
enum BpcSrc { kSrcPx = 0, kSrcPy = 1, kSrcPz = 2, kSrcVx = 3, kSrcVy = 4, kSrcVz = 5, kSrcDx = 6, kSrcDy = 7, kSrcDz = 8, kSrcOne = 9 };

// the three masks of 8 consecutive j's for this lane's body
template <bool SELF, bool PLANAR>
__device__ __forceinline__ void bpc_produce(const BpcTile &t, int jt, uint32_t jglobal, uint32_t gn, const float4 pn, const float4 vn,
                                            float r1, float t2, float t3, BpcRing &ring, int j0, int lane)
{
    float pxs[kBpcPer], pys[kBpcPer], pzs[kBpcPer], vxs[kBpcPer], vys[kBpcPer], vzs[kBpcPer];
#pragma unroll
    for (int q = 0; q < kBpcPer / 4; ++q) {
        const float4 px = *reinterpret_cast<const float4 *>(&t.c[kSrcPx][jt + 4 * q]);
        const float4 py = *reinterpret_cast<const float4 *>(&t.c[kSrcPy][jt + 4 * q]);
        const float4 vx = *reinterpret_cast<const float4 *>(&t.c[kSrcVx][jt + 4 * q]);
        const float4 vy = *reinterpret_cast<const float4 *>(&t.c[kSrcVy][jt + 4 * q]);
        float4 pz = make_float4(0.f, 0.f, 0.f, 0.f), vz = pz;
        if (!PLANAR) {
            pz = *reinterpret_cast<const float4 *>(&t.c[kSrcPz][jt + 4 * q]);
            vz = *reinterpret_cast<const float4 *>(&t.c[kSrcVz][jt + 4 * q]);
        }
        pxs[4 * q] = px.x, pxs[4 * q + 1] = px.y, pxs[4 * q + 2] = px.z, pxs[4 * q + 3] = px.w;
        pys[4 * q] = py.x, pys[4 * q + 1] = py.y, pys[4 * q + 2] = py.z, pys[4 * q + 3] = py.w;
        pzs[4 * q] = pz.x, pzs[4 * q + 1] = pz.y, pzs[4 * q + 2] = pz.z, pzs[4 * q + 3] = pz.w;
        vxs[4 * q] = vx.x, vxs[4 * q + 1] = vx.y, vxs[4 * q + 2] = vx.z, vxs[4 * q + 3] = vx.w;
        vys[4 * q] = vy.x, vys[4 * q + 1] = vy.y, vys[4 * q + 2] = vy.z, vys[4 * q + 3] = vy.w;
        vzs[4 * q] = vz.x, vzs[4 * q + 1] = vz.y, vzs[4 * q + 2] = vz.z, vzs[4 * q + 3] = vz.w;
    }
    half8 m1, m2, m3;
#pragma unroll
    for (int u = 0; u < kBpcPer; ++u) {
        const bool ne = SELF ? (jglobal + (uint32_t)u != gn) : true;          // main.rs:475 n != i
        const float dx = pxs[u] - pn.x, dy = pys[u] - pn.y;
        float d2 = (dx * dx) + (dy * dy);
        if (!PLANAR) {
            const float dz = pzs[u] - pn.z;
            d2 = d2 + (dz * dz);
        }
        const float ex = vxs[u] - vn.x, ey = vys[u] - vn.y;
        float e2 = (ex * ex) + (ey * ey);
        if (!PLANAR) {
            const float ez = vzs[u] - vn.z;
            e2 = e2 + (ez * ez);
        }
        m1[u] = ((d2 < r1) && ne) ? (_Float16)1.0f : (_Float16)0.0f;    // main.rs:474-475
        m2[u] = ((d2 <= t2) && ne) ? (_Float16)-1.0f : (_Float16)0.0f;  // main.rs:485-486 (negated: repel SUBTRACTS)
        m3[u] = ((e2 <= t3) && ne) ? (_Float16)1.0f : (_Float16)0.0f;   // main.rs:497-498
    }
    *reinterpret_cast<half8 *>(&ring.m[0][lane][j0]) = m1;
    *reinterpret_cast<half8 *>(&ring.m[1][lane][j0]) = m2;
    *reinterpret_cast<half8 *>(&ring.m[2][lane][j0]) = m3;
}

// One chain over one chunk: sum = sum (+) mask_j * src_j for j = 0..nvalid-1 in order.  MASKED: fma form (finite data);
// else select-on-result.  src: a tile plane, a difference to the own coordinate (repel), or the constant 1 (counts).
// 16 j's per trip: two 16-byte mask reads (8 halfs each) and four 16-byte source reads.
template <int SRC, bool MASKED>
__device__ __forceinline__ void bpc_chain(const _Float16 (*mrow)[kBpcStride], const BpcTile &t, int jt, float own, int lane, int nvalid,
                                          float &sum)
{
    constexpr int plane = SRC == kSrcOne ? 0 : SRC >= kSrcDx ? SRC - kSrcDx : SRC;
#pragma unroll 1
    for (int h = 0; h < kBpcChunk / 16; ++h) {
        half8 mk[2];
        float4 sv[4];
#pragma unroll
        for (int g = 0; g < 2; ++g) mk[g] = *reinterpret_cast<const half8 *>(&mrow[lane][16 * h + 8 * g]);
#pragma unroll
        for (int g = 0; g < 4; ++g)
            sv[g] = SRC != kSrcOne ? *reinterpret_cast<const float4 *>(&t.c[plane][jt + 16 * h + 4 * g]) : make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (16 * h + u >= nvalid) continue;  // uniform; only the last chunk of the set is short
            const _Float16 mh = mk[u / 8][u % 8];
            const float4 v4 = sv[u / 4];
            float x = (u % 4) == 0 ? v4.x : (u % 4) == 1 ? v4.y : (u % 4) == 2 ? v4.z : v4.w;
            if (SRC >= kSrcDx && SRC <= kSrcDz) x = x - own;  // p_i - p_n
            if (MASKED) {
                sum = SRC == kSrcOne ? sum + (float)mh : __builtin_fmaf((float)mh, x, sum);
            } else {  // select on the result: a skipped term leaves the sum untouched whatever x is
                const float cand = (SRC >= kSrcDx && SRC <= kSrcDz) ? sum - x : sum + x;
                sum = (mh != (_Float16)0.0f) ? cand : sum;
            }
        }
    }
}

// the chain(s) of consumer wave ROLE: sum_a always, sum_b a z chain in 3-D for roles 2, 3, 7
template <int ROLE, bool MASKED, bool PLANAR>
__device__ __forceinline__ void bpc_fold(const BpcRing &rg, const BpcTile &tl, int jt, const float4 pn, int lane, int nvalid, float &sum_a,
                                         float &sum_b)
{
    if (ROLE == 0) bpc_chain<kSrcPx, MASKED>(rg.m[0], tl, jt, 0.f, lane, nvalid, sum_a);                     // centre x
    if (ROLE == 1) bpc_chain<kSrcPy, MASKED>(rg.m[0], tl, jt, 0.f, lane, nvalid, sum_a);                     // centre y
    if (ROLE == 2) {
        bpc_chain<kSrcOne, MASKED>(rg.m[0], tl, jt, 0.f, lane, nvalid, sum_a);                               // count
        if (!PLANAR) bpc_chain<kSrcPz, MASKED>(rg.m[0], tl, jt, 0.f, lane, nvalid, sum_b);                   // centre z
    }
    if (ROLE == 3) {
        bpc_chain<kSrcDx, MASKED>(rg.m[1], tl, jt, pn.x, lane, nvalid, sum_a);                               // repel x
        if (!PLANAR) bpc_chain<kSrcDz, MASKED>(rg.m[1], tl, jt, pn.z, lane, nvalid, sum_b);                  // repel z
    }
    if (ROLE == 4) bpc_chain<kSrcDy, MASKED>(rg.m[1], tl, jt, pn.y, lane, nvalid, sum_a);                    // repel y
    if (ROLE == 5) bpc_chain<kSrcVx, MASKED>(rg.m[2], tl, jt, 0.f, lane, nvalid, sum_a);                     // match x
    if (ROLE == 6) bpc_chain<kSrcVy, MASKED>(rg.m[2], tl, jt, 0.f, lane, nvalid, sum_a);                     // match y
    if (ROLE == 7) {
        bpc_chain<kSrcOne, MASKED>(rg.m[2], tl, jt, 0.f, lane, nvalid, sum_a);                               // vcount
        if (!PLANAR) bpc_chain<kSrcVz, MASKED>(rg.m[2], tl, jt, 0.f, lane, nvalid, sum_b);                   // match z
    }
}

// form (masked FMA / select, planar / 3-D) chosen once per call, outside the chunk loop
template <int ROLE>
__device__ __forceinline__ void bpc_fold_flags(uint32_t flags, const BpcRing &rg, const BpcTile &tl, int jt, const float4 pn, int lane,
                                               int nvalid, float &sum_a, float &sum_b)
{
    const bool masked = (flags & kBoidsNonFinite) == 0u, planar = (flags & kBoidsNonPlanar) == 0u;
    if (masked && planar)
        bpc_fold<ROLE, true, true>(rg, tl, jt, pn, lane, nvalid, sum_a, sum_b);
    else if (masked)
        bpc_fold<ROLE, true, false>(rg, tl, jt, pn, lane, nvalid, sum_a, sum_b);
    else
        bpc_fold<ROLE, false, false>(rg, tl, jt, pn, lane, nvalid, sum_a, sum_b);
}

__global__ __launch_bounds__(kBpcThreads) void boids_pc_kernel(BoidsArgs a)
{
    __shared__ __attribute__((aligned(16))) BpcTile tile[2];
    __shared__ __attribute__((aligned(16))) BpcRing ring[2];
    __shared__ uint32_t tile_flags[2][8];
    // the 11 final sums per body are handed to wave 0 through ring[0]'s storage once every fold is done (keeps the
    // workgroup under 80 KiB of LDS, i.e. two workgroups per CU)
    float(*sums)[64] = reinterpret_cast<float(*)[64]>(&ring[0]);

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const uint32_t l = blockIdx.x * 64u + (uint32_t)lane;
    const bool live = l < a.count;
    const uint32_t gn = a.first + (live ? l : a.count - 1u);
    const float4 pn = a.pos_in[gn];
    const float4 vn = a.vel_in[gn];
    auto wave_flags = [](uint32_t f) {
        return (__any((int)(f & kBoidsNonFinite)) ? kBoidsNonFinite : 0u) | (__any((int)(f & kBoidsNonPlanar)) ? kBoidsNonPlanar : 0u);
    };
    const uint32_t self = wave_flags(boids_flags(pn, vn) | a.force_flags);  // same 64 bodies in every wave
    const uint32_t own_lo = a.first + blockIdx.x * 64u, own_hi = own_lo + 64u;

    const uint32_t n = a.n_total;
    const uint32_t ntiles = (n + (uint32_t)kBpcTile - 1u) / (uint32_t)kBpcTile;
    // loaders: the 512 threads of the producer waves stage one position and one velocity record each
    const int ltid = tid - kBpcConsumers * 64;
    const bool loader = ltid >= 0;
    struct Staged {
        float4 p, v;
    };
    auto fetch = [&](uint32_t t) -> Staged {
        Staged r = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
        const uint32_t j = t * (uint32_t)kBpcTile + (uint32_t)(loader ? ltid : 0);
        if (loader && j < n) {
            r.p = a.pos_in[j];
            r.v = a.vel_in[j];
        }
        return r;
    };
    auto publish = [&](int buf, Staged rec) {
        // keep the flag arithmetic (pure, on prefetched data) at the use point instead of right behind the loads
        asm volatile("" : "+v"(rec.p.x), "+v"(rec.p.y), "+v"(rec.p.z), "+v"(rec.p.w), "+v"(rec.v.x), "+v"(rec.v.y), "+v"(rec.v.z), "+v"(rec.v.w));
        if (loader) {
            tile[buf].c[kSrcPx][ltid] = rec.p.x;
            tile[buf].c[kSrcPy][ltid] = rec.p.y;
            tile[buf].c[kSrcPz][ltid] = rec.p.z;
            tile[buf].c[kSrcVx][ltid] = rec.v.x;
            tile[buf].c[kSrcVy][ltid] = rec.v.y;
            tile[buf].c[kSrcVz][ltid] = rec.v.z;
            const uint32_t f = wave_flags(boids_flags(rec.p, rec.v));
            if (lane == 0) tile_flags[buf][wave - kBpcConsumers] = f;
        }
    };
    publish(0, fetch(0u));
    __syncthreads();

    // per-tile bookkeeping shared by both roles (all wave-uniform)
    struct TileInfo {
        int tb, nk;
        bool stage, has_self;
        uint32_t flags, j_first;
    };
    auto tile_info = [&](uint32_t t) {
        TileInfo ti;
        ti.tb = (int)(t & 1u);
        ti.stage = t + 1u < ntiles;
        uint32_t f = self;
#pragma unroll
        for (int w = 0; w < 8; ++w) f |= tile_flags[ti.tb][w];
        ti.flags = __builtin_amdgcn_readfirstlane(f);
        ti.j_first = t * (uint32_t)kBpcTile;
        const uint32_t left = n - ti.j_first;
        ti.nk = left >= (uint32_t)kBpcTile ? kBpcChunksPerTile : (int)((left + (uint32_t)kBpcChunk - 1u) / (uint32_t)kBpcChunk);
        const uint32_t tile_n = left < (uint32_t)kBpcTile ? left : (uint32_t)kBpcTile;
        ti.has_self = ti.j_first < own_hi && ti.j_first + tile_n > own_lo;
        return ti;
    };

    // ---- consumer waves: the whole step for one role, forms chosen per tile ------------------------------
    auto consumer = [&](auto role_tag) {
        constexpr int ROLE = decltype(role_tag)::value;
        float sum_a = 0.f, sum_b = 0.f;
        uint32_t prev_flags = 0u;
        int last_nk = 1;
        for (uint32_t t = 0; t < ntiles; ++t) {
            const TileInfo ti = tile_info(t);
            last_nk = ti.nk;
            const bool first_tile = t == 0u;
            auto nothing = [] {};
            // at step k fold chunk k-1; step 0: the previous tile's last chunk (tile buffer tb^1, ring parity 1)
            pc_steps(ti.nk, [&](int k, int par) {
                if (k == 0) {
                    if (!first_tile)
                        bpc_fold_flags<ROLE>(prev_flags, ring[par ^ 1], tile[ti.tb ^ 1], (kBpcChunksPerTile - 1) * kBpcChunk, pn, lane,
                                             kBpcChunk, sum_a, sum_b);
                } else {
                    bpc_fold_flags<ROLE>(ti.flags, ring[par ^ 1], tile[ti.tb], (k - 1) * kBpcChunk, pn, lane, kBpcChunk, sum_a, sum_b);
                }
            }, nothing);
            prev_flags = ti.flags;
        }
        // drain: the last chunk of the step (its tile buffer and ring slot are still intact)
        const int tb = (int)((ntiles - 1u) & 1u), k = last_nk - 1;
        const uint32_t first_j = (ntiles - 1u) * (uint32_t)kBpcTile + (uint32_t)k * (uint32_t)kBpcChunk;
        bpc_fold_flags<ROLE>(prev_flags, ring[k & 1], tile[tb], k * kBpcChunk, pn, lane, (int)(n - first_j), sum_a, sum_b);
        __syncthreads();  // every consumer is done with the ring: its storage now carries the sums (producers match this barrier)
        sums[ROLE][lane] = sum_a;
        if (ROLE == 2) sums[8][lane] = sum_b;   // centre z
        if (ROLE == 3) sums[9][lane] = sum_b;   // repel z
        if (ROLE == 7) sums[10][lane] = sum_b;  // match z
    };

    // ---- producer waves ------------------------------------------------------------------------------------
    auto producer = [&]() {
        const int j0 = (wave - kBpcConsumers) * kBpcPer;
        for (uint32_t t = 0; t < ntiles; ++t) {
            const TileInfo ti = tile_info(t);
            Staged rec = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
            if (ti.stage) rec = fetch(t + 1u);  // in flight across the chunks of this tile
            auto stage_next = [&]() {
                if (ti.stage) publish(ti.tb ^ 1, rec);  // tile t+1 goes to the buffer tile t-1 used
            };
            const bool planar = (ti.flags & kBoidsNonPlanar) == 0u;
            auto run = [&](auto self_tag, auto planar_tag) {
                pc_steps(ti.nk, [&](int k, int par) {
                    const int jt = k * kBpcChunk + j0;
                    bpc_produce<decltype(self_tag)::value, decltype(planar_tag)::value>(tile[ti.tb], jt, ti.j_first + (uint32_t)jt, gn, pn,
                                                                                        vn, a.r1, a.t2, a.t3, ring[par], j0, lane);
                }, stage_next);
            };
            using T = std::true_type;
            using F = std::false_type;
            if (ti.has_self) {
                if (planar) run(T{}, T{}); else run(T{}, F{});
            } else {
                if (planar) run(F{}, T{}); else run(F{}, F{});
            }
        }
        __syncthreads();  // matches the consumers' barrier in front of the sums hand-off
    };

    switch (wave) {
        case 0: consumer(std::integral_constant<int, 0>{}); break;
        case 1: consumer(std::integral_constant<int, 1>{}); break;
        case 2: consumer(std::integral_constant<int, 2>{}); break;
        case 3: consumer(std::integral_constant<int, 3>{}); break;
        case 4: consumer(std::integral_constant<int, 4>{}); break;
        case 5: consumer(std::integral_constant<int, 5>{}); break;
        case 6: consumer(std::integral_constant<int, 6>{}); break;
        case 7: consumer(std::integral_constant<int, 7>{}); break;
        default: producer(); break;
    }
    __syncthreads();
    if (wave == 0 && live) {
        float cx = sums[0][lane], cy = sums[1][lane], cz = sums[8][lane], mx = sums[5][lane], my = sums[6][lane], mz = sums[10][lane];
        const float cnt = sums[2][lane], vcnt = sums[7][lane], rx = sums[3][lane], ry = sums[4][lane], rz = sums[9][lane];
        if (cnt > 0.f) {  // main.rs:506-508
            cx = cx / cnt;
            cy = cy / cnt;
            cz = cz / cnt;
        }
        if (vcnt > 0.f) {  // main.rs:510-512
            mx = mx / vcnt;
            my = my / vcnt;
            mz = mz / vcnt;
        }
        const float ax = cx * a.s1, ay = cy * a.s1, az = cz * a.s1;  // main.rs:514
        const float bx = rx * a.s2, by = ry * a.s2, bz = rz * a.s2;
        const float gx = mx * a.s3, gy = my * a.s3, gz = mz * a.s3;
        float vx = (ax + bx) + gx, vy = (ay + by) + gy, vz = (az + bz) + gz;
        const float q0 = vx * vx, q1 = vy * vy, q2 = vz * vz;
        const float mag = __builtin_sqrtf((q0 + q1) + q2);  // main.rs:516-518
        if (mag > 1.0f) {
            const float sc = 1.0f / mag;
            vx = vx * sc;
            vy = vy * sc;
            vz = vz * sc;
        }
        const float sx = vx * a.dt, sy = vy * a.dt, sz = vz * a.dt;  // main.rs:521
        a.vel_out[a.first + l] = make_float4(vx, vy, vz, 0.f);
        a.pos_out[a.first + l] = make_float4(sx + pn.x, sy + pn.y, sz + pn.z, 0.f);
    }
}

hipError_t launch_boids_pc(const BoidsArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(boids_pc_kernel, dim3(ceil_div_u(a.count, 64)), dim3(kBpcThreads), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_boids(const BoidsArgs &a, uint32_t tile, uint32_t pc, hipStream_t s)
{
    if (pc) return launch_boids_pc(a, s);
    const dim3 grid(ceil_div_u(a.count, kBlock)), block(kBlock);
    switch (tile) {
        case 256: hipLaunchKernelGGL(boids_step_kernel<256>, grid, block, 0, s, a); break;
        case 512: hipLaunchKernelGGL(boids_step_kernel<512>, grid, block, 0, s, a); break;
        case 1024: hipLaunchKernelGGL(boids_step_kernel<1024>, grid, block, 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

#endif  // NBK_NOSLP_TU

#ifndef NBK_NOSLP_TU
// ------------------------------------------------------------------------------------------------
// STRICT, producer/consumer form ("pc"): the same arithmetic and the same summation order, for shards too small
// to fill the chip with one lane per body.
//
// A workgroup owns 64 bodies (lane = body) and has 2 + NP waves (NP = 8 or 14).  Waves 2.. are PRODUCERS: for a chunk of
// 4*NP j's each evaluates 4 consecutive pairs per body -- everything of main.rs:428-430 except the running-sum add, i.e. the part
// with no ordering constraint -- and writes the quotients to an LDS ring q[parity][component][body][j] (one
// ds_write_b128 per component).  Waves 0 and 1 are CONSUMERS: wave 0 owns every body's x sum (and z sum), wave 1
// the y sum; each walks the previous chunk's ring in j order doing `sum = sum + q` (main.rs:430), four j's per
// ds_read_b128, so every component of every body's fold sees exactly the reference's sequence of additions.
// One barrier per chunk; the ring is double buffered, so producers of chunk c overlap the consumers of chunk c-1.
// Positions are staged in 256-record tiles one tile ahead (waves 0..3 load).  VALU work per pair equals the
// one-lane-per-body kernel (the adds just run in other waves); the wave count is 10 per 64 bodies instead of 1.
// Why two consumers: a wave alone issues one instruction per ~5 cycles, and the chain needs 1 add per j per
// component, so a single consumer for both components would cap the step at N x 15 cycles.
// ------------------------------------------------------------------------------------------------
constexpr int kPcConsumers = 2;
constexpr int kPcPer = 4;  // consecutive j's per producer per chunk

// Shape of one producer/consumer workgroup, by producer count NP (8 or 14)
template <int NP>
struct PcShape {
    static constexpr int kWaves = kPcConsumers + NP;
    static constexpr int kThreads = kWaves * 64;
    static constexpr int kChunk = NP * kPcPer;        // j's per chunk: 32 / 56
    static constexpr int kStride = kChunk + 4;        // floats per body row: rows stay 16-B aligned, banks spread
    static constexpr int kChunksPerTile = 8;
    static constexpr int kTile = kChunk * kChunksPerTile;  // position records per staging tile: 256 / 448
    static constexpr int kLoaderWaves = kTile / 64;
    static_assert(kTile % 64 == 0 && kTile <= kThreads, "tile is loaded one record per thread");
    struct Ring {
        float q[3][64][kStride];  // [component][body][j within chunk]
    };
};

template <int NP, bool IEEE, bool PLANAR>
__device__ __forceinline__ void pc_produce(const float4 *chunk, int j0, int lane, float xi, float yi, float zi, float G,
                                           typename PcShape<NP>::Ring &ring)
{
    float qx[kPcPer], qy[kPcPer], qz[kPcPer];
#pragma unroll
    for (int t = 0; t < kPcPer; ++t) pair_strict<IEEE, PLANAR>(chunk[j0 + t], xi, yi, zi, G, qx[t], qy[t], qz[t]);
    // records past the end of the set are zero-padded by the loader; whatever they produce is never added
    *reinterpret_cast<float4 *>(&ring.q[0][lane][j0]) = make_float4(qx[0], qx[1], qx[2], qx[3]);
    *reinterpret_cast<float4 *>(&ring.q[1][lane][j0]) = make_float4(qy[0], qy[1], qy[2], qy[3]);
    if (!PLANAR) *reinterpret_cast<float4 *>(&ring.q[2][lane][j0]) = make_float4(qz[0], qz[1], qz[2], qz[3]);
}

// sum = sum + q[j] for j = 0 .. nvalid-1 in order: main.rs:430 for one component of one body per lane.
// Read in two halves (at most 7 float4 = 28 registers in flight) so the consumer role does not set the kernel's
// register count.
template <int NP>
__device__ __forceinline__ void pc_consume(const float (*row)[PcShape<NP>::kStride], int lane, int nvalid, float &sum)
{
    constexpr int kChunk = PcShape<NP>::kChunk;
    constexpr int kHalf = kChunk / 8;  // float4 per half: 4 (NP = 8) or 7 (NP = 14)
    const float *base = &row[lane][0];
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        float4 r[kHalf];
#pragma unroll
        for (int g = 0; g < kHalf; ++g) r[g] = *reinterpret_cast<const float4 *>(base + 4 * (h * kHalf + g));
        const int j0 = 4 * h * kHalf;
        if (nvalid == kChunk) {
#pragma unroll
            for (int g = 0; g < kHalf; ++g) {
                sum = sum + r[g].x;
                sum = sum + r[g].y;
                sum = sum + r[g].z;
                sum = sum + r[g].w;
            }
        } else {  // last chunk of the set
#pragma unroll
            for (int g = 0; g < kHalf; ++g) {
                if (j0 + 4 * g + 0 < nvalid) sum = sum + r[g].x;
                if (j0 + 4 * g + 1 < nvalid) sum = sum + r[g].y;
                if (j0 + 4 * g + 2 < nvalid) sum = sum + r[g].z;
                if (j0 + 4 * g + 3 < nvalid) sum = sum + r[g].w;
            }
        }
    }
}

template <int NP>
__global__ __launch_bounds__(PcShape<NP>::kThreads) void step_strict_pc_kernel(StepArgs a)
{
    using Sh = PcShape<NP>;
    constexpr int kChunk = Sh::kChunk, kTile = Sh::kTile, kChunksPerTile = Sh::kChunksPerTile;
    __shared__ float4 tile[2][kTile];
    __shared__ __attribute__((aligned(16))) typename Sh::Ring ring[2];
    __shared__ uint32_t tile_flags[2][Sh::kLoaderWaves];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: role branches become scalar
    const int lane = tid & 63;
    const uint32_t l = blockIdx.x * 64u + (uint32_t)lane;  // body index inside the shard: the same 64 bodies in every wave
    const bool live = l < a.count;
    const float4 pi = a.pos_in[a.first + (live ? l : a.count - 1u)];
    const float xi = pi.x, yi = pi.y, zi = pi.z;

    const uint32_t lo = a.lo_bits, span = a.hi_bits - a.lo_bits;
    // every wave holds the same 64 bodies, so each derives the workgroup's own-body flags by itself
    const uint32_t self = wave_or(coord_oor(xi, lo, span) | coord_oor(yi, lo, span) | coord_oor(zi, lo, span) | a.force_ieee |
                                  nonzero_bits(zi) | a.force_3d);

    const uint32_t n = a.n_total;
    const uint32_t ntiles = (n + (uint32_t)kTile - 1u) / (uint32_t)kTile;
    const uint32_t nfull = n / (uint32_t)kTile;  // tiles whose 8 chunks are all complete
    const bool loader = wave < Sh::kLoaderWaves;
    auto fetch = [&](uint32_t t) -> float4 {
        const uint32_t j = t * (uint32_t)kTile + (uint32_t)tid;
        return (loader && j < n) ? a.pos_in[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto publish = [&](int buf, float4 rec) {
        // Opaque use point: keeps the range/planarity checks (pure arithmetic on the prefetched record) from being
        // hoisted above the chunk loop, which would put the s_waitcnt for the global load right behind the load.
        // .w too: keeps the whole destination register range of the dwordx4 load reserved until here.
        asm volatile("" : "+v"(rec.x), "+v"(rec.y), "+v"(rec.z), "+v"(rec.w));
        if (loader) {
            const uint32_t f = wave_or(coord_oor(rec.x, lo, span) | coord_oor(rec.y, lo, span) | coord_oor(rec.z, lo, span) |
                                       nonzero_bits(rec.z));
            tile[buf][tid] = make_float4(rec.x, rec.y, rec.z, a.bias);
            if (lane == 0) tile_flags[buf][wave] = f;
        }
    };
    publish(0, fetch(0u));
    __syncthreads();

    float sum0 = 0.f, sum2 = 0.f;  // main.rs:426.  wave 0: sum0 = x, sum2 = z;  wave 1: sum0 = y
    uint32_t prev_flags = 0u;      // flags of the previous tile (the consumers fold its last chunk at this tile's chunk 0)
    uint32_t last_nk = 0u;
    for (uint32_t t = 0; t < ntiles; ++t) {
        const int tb = (int)(t & 1u);
        const bool stage = t + 1u < ntiles;
        float4 rec = make_float4(0.f, 0.f, 0.f, 0.f);
        if (stage) rec = fetch(t + 1u);  // global loads stay in flight across the chunks of this tile
        uint32_t cur_flags = self;
#pragma unroll
        for (int w = 0; w < Sh::kLoaderWaves; ++w) cur_flags |= tile_flags[tb][w];
        cur_flags = __builtin_amdgcn_readfirstlane(cur_flags);
        const uint32_t left = n - t * (uint32_t)kTile;
        const int nk = t < nfull ? kChunksPerTile : (int)((left + (uint32_t)kChunk - 1u) / (uint32_t)kChunk);
        last_nk = (uint32_t)nk;
        // Tile t+1 goes to the buffer tile t-1 used: its readers finished at least one barrier ago.
        auto stage_next = [&]() {
            if (stage) publish(tb ^ 1, rec);
        };
        const bool use_z = (cur_flags & (kFlagNonPlanar | kFlagIeee)) != 0u;
        if (wave < kPcConsumers) {
            // at step k fold chunk k-1 (step 0: the previous tile's last chunk, ring parity 1; nothing on the first tile)
            const bool z_prev = wave == 0 && (prev_flags & (kFlagNonPlanar | kFlagIeee)) != 0u;
            const bool z_this = wave == 0 && use_z;
            const bool first_tile = t == 0u;
            pc_steps(nk, [&](int k, int par) {
                if (k == 0 && first_tile) return;
                pc_consume<NP>(ring[par ^ 1].q[wave], lane, kChunk, sum0);
                if (k == 0 ? z_prev : z_this) pc_consume<NP>(ring[par ^ 1].q[2], lane, kChunk, sum2);
            }, stage_next);
        } else {
            const int j0 = (wave - kPcConsumers) * kPcPer;
            const float4 *tl = tile[tb];
            if (cur_flags == 0u)
                pc_steps(nk, [&](int k, int par) { pc_produce<NP, false, true>(tl + k * kChunk, j0, lane, xi, yi, zi, a.G, ring[par]); },
                         stage_next);
            else if ((cur_flags & kFlagIeee) == 0u)
                pc_steps(nk, [&](int k, int par) { pc_produce<NP, false, false>(tl + k * kChunk, j0, lane, xi, yi, zi, a.G, ring[par]); },
                         stage_next);
            else
                pc_steps(nk, [&](int k, int par) { pc_produce<NP, true, false>(tl + k * kChunk, j0, lane, xi, yi, zi, a.G, ring[par]); },
                         stage_next);
        }
        prev_flags = cur_flags;
    }

    // drain: the last chunk of the step (ordered after its producers by the last barrier above)
    if (wave < kPcConsumers) {
        const uint32_t last_chunk_first = (ntiles - 1u) * (uint32_t)kTile + (last_nk - 1u) * (uint32_t)kChunk;
        const int nvalid = (int)(n - last_chunk_first);  // 1 .. kChunk
        const int rb = (int)((last_nk - 1u) & 1u);      // ring parity of chunk k within its tile is k & 1
        pc_consume<NP>(ring[rb].q[wave], lane, nvalid, sum0);
        if (wave == 0 && (prev_flags & (kFlagNonPlanar | kFlagIeee)) != 0u) pc_consume<NP>(ring[rb].q[2], lane, nvalid, sum2);
    }
    __syncthreads();
    // the y sums live in wave 1: hand them to wave 0 through the ring (all folds are done)
    if (wave == 1) ring[0].q[1][lane][0] = sum0;
    __syncthreads();
    if (wave == 0 && live) {
        const float sy = ring[0].q[1][lane][0];
        float4 p = pi;
        float4 v = a.vel[l];
        integrate(p, v, sum0, sy, sum2, a.dt);
        a.vel[l] = v;
        a.pos_out[a.first + l] = p;
    }
}

// ------------------------------------------------------------------------------------------------
// FAST arithmetic
// ------------------------------------------------------------------------------------------------
template <int IB, int U, bool PLANAR>
__device__ __forceinline__ void fold_tile_fast(const float4 *tile, int nj, const float (&xi)[IB], const float (&yi)[IB],
                                               const float (&zi)[IB], float (&ax)[IB], float (&ay)[IB], float (&az)[IB])
{
    auto pair = [&](const float4 pj) {
#pragma unroll
        for (int b = 0; b < IB; ++b) {
            const float dx = pj.x - xi[b], dy = pj.y - yi[b];
            float r2 = __builtin_fmaf(dx, dx, pj.w);  // pj.w holds bias
            r2 = __builtin_fmaf(dy, dy, r2);
            float dz = 0.f;
            if (!PLANAR) {
                dz = pj.z - zi[b];
                r2 = __builtin_fmaf(dz, dz, r2);
            }
            const float inv = __builtin_amdgcn_rcpf(r2);
            ax[b] = __builtin_fmaf(dx, inv, ax[b]);
            ay[b] = __builtin_fmaf(dy, inv, ay[b]);
            if (!PLANAR) az[b] = __builtin_fmaf(dz, inv, az[b]);
        }
    };
    int j = 0;
    for (; j + U <= nj; j += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) pair(tile[j + u]);
    }
    for (; j < nj; ++j) pair(tile[j]);
}

// grid.x: blocks of 256*IB bodies; grid.y: slices of the j range (a.j_chunk records each, a multiple of TJ).
// gridDim.y == 1: fused integrate epilogue.  Otherwise partial sums go to a.partial[jy][count] and
// integrate_partials_kernel combines them in slice order.
template <int TJ, int IB>
__global__ __launch_bounds__(kBlock) void step_fast_kernel(StepArgs a)
{
    __shared__ float4 tile[2][TJ];
    __shared__ uint32_t tile_flags[2][kWaves];
    __shared__ uint32_t self_flags[kWaves];

    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const uint32_t base = blockIdx.x * (uint32_t)(kBlock * IB) + (uint32_t)tid;
    float xi[IB], yi[IB], zi[IB], ax[IB], ay[IB], az[IB];
    uint32_t sf = a.force_3d;
#pragma unroll
    for (int b = 0; b < IB; ++b) {
        const uint32_t l = base + (uint32_t)(b * kBlock);
        const float4 p = a.pos_in[a.first + (l < a.count ? l : a.count - 1u)];
        xi[b] = p.x;
        yi[b] = p.y;
        zi[b] = p.z;
        ax[b] = ay[b] = az[b] = 0.f;
        sf |= nonzero_bits(p.z);
    }
    sf = wave_or(sf);
    if ((tid & 63) == 0) self_flags[wave] = sf;

    const uint32_t j_lo = blockIdx.y * a.j_chunk;
    const uint32_t j_hi = (j_lo + a.j_chunk < a.n_total) ? j_lo + a.j_chunk : a.n_total;
    const uint32_t span = j_hi > j_lo ? j_hi - j_lo : 0u;
    const uint32_t ntiles = (span + (uint32_t)TJ - 1u) / (uint32_t)TJ;

    TileRegs<TJ> regs;
    auto publish = [&](int buf) {
        uint32_t f = 0;
#pragma unroll
        for (int k = 0; k < TJ / kBlock; ++k) f |= nonzero_bits(regs.r[k].z);
        f = wave_or(f);
        tile_store<TJ>(regs, tile[buf], tid, a.bias);
        if ((tid & 63) == 0) tile_flags[buf][wave] = f;
    };
    if (ntiles) {
        tile_fetch<TJ>(regs, a.pos_in, j_lo, j_hi, tid);
        publish(0);
    }
    __syncthreads();
    const uint32_t block_self = self_flags[0] | self_flags[1] | self_flags[2] | self_flags[3];
    int buf = 0;
    for (uint32_t t = 0; t < ntiles; ++t) {
        const bool more = (t + 1u) < ntiles;
        if (more) tile_fetch<TJ>(regs, a.pos_in, j_lo + (t + 1u) * (uint32_t)TJ, j_hi, tid);
        const uint32_t left = span - t * (uint32_t)TJ;
        const int nj = left < (uint32_t)TJ ? (int)left : TJ;
        const uint32_t f = block_self | tile_flags[buf][0] | tile_flags[buf][1] | tile_flags[buf][2] | tile_flags[buf][3];
        if (f == 0u)
            fold_tile_fast<IB, 8 / IB, true>(tile[buf], nj, xi, yi, zi, ax, ay, az);
        else
            fold_tile_fast<IB, 8 / IB, false>(tile[buf], nj, xi, yi, zi, ax, ay, az);
        if (more) publish(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

#pragma unroll
    for (int b = 0; b < IB; ++b) {
        const uint32_t l = base + (uint32_t)(b * kBlock);
        if (l >= a.count) continue;
        if (gridDim.y == 1) {
            float4 p = make_float4(xi[b], yi[b], zi[b], 0.f);
            float4 v = a.vel[l];
            integrate(p, v, ax[b] * a.G, ay[b] * a.G, az[b] * a.G, a.dt);
            a.vel[l] = v;
            a.pos_out[a.first + l] = p;
        } else {
            a.partial[(size_t)blockIdx.y * a.count + l] = make_float4(ax[b], ay[b], az[b], 0.f);
        }
    }
}

__global__ __launch_bounds__(kBlock) void integrate_partials_kernel(StepArgs a, uint32_t slices)
{
    const uint32_t l = blockIdx.x * (uint32_t)kBlock + threadIdx.x;
    if (l >= a.count) return;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (uint32_t s = 0; s < slices; ++s) {  // fixed order: deterministic
        const float4 q = a.partial[(size_t)s * a.count + l];
        sx = sx + q.x;
        sy = sy + q.y;
        sz = sz + q.z;
    }
    float4 p = a.pos_in[a.first + l];
    float4 v = a.vel[l];
    integrate(p, v, sx * a.G, sy * a.G, sz * a.G, a.dt);
    a.vel[l] = v;
    a.pos_out[a.first + l] = p;
}

// ------------------------------------------------------------------------------------------------
// model matrices, main.rs:437-439:  M = from_translation(p) * from_angle_z(atan2(v.y, v.x)), evaluated as
// the explicit column-by-column product cgmath performs (so signed zeros and non-finite positions come
// out the way they do in the reference).  atan2f/sinf/cosf are the device libm's: within a few ulp of
// the host libm the reference uses; tests hold them to 1e-6 absolute.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void instances_kernel(uint32_t count, const float4 *__restrict__ pos,
                                                           const float4 *__restrict__ vel, float4 *__restrict__ inst)
{
    const uint32_t l = blockIdx.x * (uint32_t)kBlock + threadIdx.x;
    if (l >= count) return;
    const float4 p = pos[l];
    const float4 v = vel[l];
    const float theta = atan2f(v.y, v.x);
    const float s = sinf(theta);
    const float c = cosf(theta);
    const float A[4] = {1.f, 0.f, 0.f, 0.f}, B[4] = {0.f, 1.f, 0.f, 0.f}, C[4] = {0.f, 0.f, 1.f, 0.f};
    const float D[4] = {p.x, p.y, p.z, 1.f};
    const float R[4][4] = {{c, s, 0.f, 0.f}, {-s, c, 0.f, 0.f}, {0.f, 0.f, 1.f, 0.f}, {0.f, 0.f, 0.f, 1.f}};
    float m[16];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float t0 = A[e] * R[k][0], t1 = B[e] * R[k][1], t2 = C[e] * R[k][2], t3 = D[e] * R[k][3];
            m[4 * k + e] = ((t0 + t1) + t2) + t3;
        }
    float4 *out = inst + (size_t)l * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = make_float4(m[4 * k], m[4 * k + 1], m[4 * k + 2], m[4 * k + 3]);
}

// ------------------------------------------------------------------------------------------------
// CameraArray::update, gfx.rs:397-408 / build_camera gfx.rs:358-369: one thread per
// entity, M = CP * look_at_dir(eye, dir, up) with CP = correction * proj passed in by the caller (a constant of the
// array).  normalize() = self * (1 / magnitude) with the correctly rounded sqrt and '/'.  Bit-identical to the oracle.
// ------------------------------------------------------------------------------------------------
struct Mat4Arg {
    float m[16];  // column-major
};

__device__ __forceinline__ void normalize3(float &x, float &y, float &z)
{
    const float q0 = x * x, q1 = y * y, q2 = z * z;
    const float mag = __builtin_sqrtf((q0 + q1) + q2);
    const float s = 1.0f / mag;
    x = x * s;
    y = y * s;
    z = z * s;
}

__global__ __launch_bounds__(kBlock) void cameras_kernel(uint32_t count, const float4 *__restrict__ eyes,
                                                         const float4 *__restrict__ dirs, float ux, float uy, float uz, Mat4Arg cp,
                                                         float4 *__restrict__ out)
{
    const uint32_t l = blockIdx.x * (uint32_t)kBlock + threadIdx.x;
    if (l >= count) return;
    const float4 e = eyes[l];
    const float4 d = dirs[l];
    float fx = d.x, fy = d.y, fz = d.z;
    normalize3(fx, fy, fz);                                   // f = dir.normalize()
    const float a0 = fy * uz, a1 = fz * uy, b0 = fz * ux, b1 = fx * uz, c0 = fx * uy, c1 = fy * ux;
    float sx = a0 - a1, sy = b0 - b1, sz = c0 - c1;           // s = f.cross(up)
    normalize3(sx, sy, sz);
    const float g0 = sy * fz, g1 = sz * fy, h0 = sz * fx, h1 = sx * fz, i0 = sx * fy, i1 = sy * fx;
    const float wx = g0 - g1, wy = h0 - h1, wz = i0 - i1;     // u = s.cross(f)
    const float es0 = e.x * sx, es1 = e.y * sy, es2 = e.z * sz;
    const float eu0 = e.x * wx, eu1 = e.y * wy, eu2 = e.z * wz;
    const float ef0 = e.x * fx, ef1 = e.y * fy, ef2 = e.z * fz;
    const float es = (es0 + es1) + es2, eu = (eu0 + eu1) + eu2, ef = (ef0 + ef1) + ef2;
    const float view[4][4] = {{sx, wx, -fx, 0.f}, {sy, wy, -fy, 0.f}, {sz, wz, -fz, 0.f}, {-es, -eu, ef, 1.f}};
    float m[16];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float t0 = cp.m[0 + r] * view[k][0], t1 = cp.m[4 + r] * view[k][1];
            const float t2 = cp.m[8 + r] * view[k][2], t3 = cp.m[12 + r] * view[k][3];
            m[4 * k + r] = ((t0 + t1) + t2) + t3;
        }
    float4 *o = out + (size_t)l * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = make_float4(m[4 * k], m[4 * k + 1], m[4 * k + 2], m[4 * k + 3]);
}

hipError_t launch_cameras(uint32_t count, const float4 *eyes, const float4 *dirs, const float *up3, const float *cp16, float4 *out,
                          hipStream_t s)
{
    Mat4Arg cp;
    for (int i = 0; i < 16; ++i) cp.m[i] = cp16[i];
    hipLaunchKernelGGL(cameras_kernel, dim3(ceil_div_u(count, kBlock)), dim3(kBlock), 0, s, count, eyes, dirs, up3[0], up3[1], up3[2],
                       cp, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Random-walk controller, update_instance_random main.rs:381-402: in place, O(N).
// The jitter stream is build-owned and counter based (seed, step, body): see the oracle's random_jitter.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float random_jitter(uint64_t seed, uint64_t step, uint32_t body, uint32_t which)
{
    uint64_t s = seed ^ (step * 0xD1B54A32D192ED03ull) ^ ((uint64_t)body * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)which << 62);
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    const float u = (float)(uint32_t)(z >> 40) * 0x1.0p-24f;
    const float lo = -0.0001f, hi = 0.0001f;
    const float scale = hi - lo;
    const float v = scale * u;
    return lo + v;
}

__global__ __launch_bounds__(kBlock) void random_step_kernel(uint32_t first, uint32_t count, float4 *__restrict__ pos,
                                                             float4 *__restrict__ vel, uint64_t seed, uint64_t step)
{
    const uint32_t l = blockIdx.x * (uint32_t)kBlock + threadIdx.x;
    if (l >= count) return;
    float4 p = pos[l], v = vel[l];
    const float jx = random_jitter(seed, step, first + l, 0), jy = random_jitter(seed, step, first + l, 1);
    v.x = v.x + jx;  // main.rs:392-396
    v.y = v.y + jy;
    v.z = v.z + 0.0f;
    p.x = p.x + v.x;  // main.rs:397
    p.y = p.y + v.y;
    p.z = p.z + v.z;
    pos[l] = p;
    vel[l] = v;
}

hipError_t launch_random(uint32_t first, uint32_t count, float4 *pos, float4 *vel, uint64_t seed, uint64_t step, hipStream_t s)
{
    hipLaunchKernelGGL(random_step_kernel, dim3(ceil_div_u(count, kBlock)), dim3(kBlock), 0, s, first, count, pos, vel, seed, step);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Self-test of the STRICT division (diagnostic; nb_selftest_divide): random numerators and denominators covering the
// whole exponent rectangle the range guard admits, the shared-reciprocal ladder against the IEEE '/'.
// Every thread draws `per_thread` pairs: mantissas uniform (plus all-zeros / all-ones / one-bit patterns every 8th
// draw), exponents uniform in [d_lo, d_hi] x [n_lo, n_hi], random numerator sign.  Mismatching bit patterns are counted.
// ------------------------------------------------------------------------------------------------
template <bool SINGLE_CORRECTION>  // true = control arm (no correction step at all)
__global__ __launch_bounds__(kBlock) void divide_selftest_kernel(uint64_t seed, uint32_t per_thread, int d_lo, int d_hi, int n_lo,
                                                                 int n_hi, unsigned long long *mismatches, float *first_bad)
{
    uint64_t s = seed ^ ((uint64_t)(blockIdx.x * (uint32_t)kBlock + threadIdx.x) * 0x9E3779B97F4A7C15ull);
    auto next = [&]() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    unsigned long long bad = 0;
    for (uint32_t i = 0; i < per_thread; ++i) {
        const uint64_t a = next(), b = next();
        uint32_t md = (uint32_t)(a & 0x7fffffu), mn = (uint32_t)(b & 0x7fffffu);
        if ((i & 7u) == 7u) {  // structured mantissas: the reciprocal refinement's hard cases live at the ends
            const uint32_t sel = (uint32_t)(a >> 23) & 3u;
            md = sel == 0 ? 0u : sel == 1 ? 0x7fffffu : sel == 2 ? (1u << ((a >> 25) % 23)) : 0x7fffffu ^ (1u << ((a >> 25) % 23));
            const uint32_t seln = (uint32_t)(b >> 23) & 3u;
            mn = seln == 0 ? 0u : seln == 1 ? 0x7fffffu : mn;
        }
        const int ed = d_lo + (int)((a >> 32) % (uint64_t)(d_hi - d_lo + 1));
        const int en = n_lo + (int)((b >> 32) % (uint64_t)(n_hi - n_lo + 1));
        const float d = __uint_as_float(((uint32_t)(ed + 127) << 23) | md);
        const float n = __uint_as_float(((uint32_t)(b >> 63) << 31) | ((uint32_t)(en + 127) << 23) | mn);
        const float r0 = __builtin_amdgcn_rcpf(d);
        const float e = __builtin_fmaf(-d, r0, 1.0f);
        const float r = __builtin_fmaf(e, r0, r0);
        float q_ladder;
        if (SINGLE_CORRECTION) {  // control arm: the uncorrected product n * r, off by an ulp in a few percent of the draws
            q_ladder = n * r;
        } else {
            q_ladder = div_ladder(n, d, r);
        }
        const float q_ieee = n / d;
        if (__float_as_uint(q_ladder) != __float_as_uint(q_ieee)) {
            if (bad == 0 && first_bad) {
                first_bad[0] = n;
                first_bad[1] = d;
            }
            ++bad;
        }
    }
    if (bad) atomicAdd(mismatches, bad);
}

hipError_t launch_divide_selftest(uint32_t blocks, uint64_t seed, uint32_t per_thread, int d_lo, int d_hi, int n_lo, int n_hi,
                                  unsigned long long *mismatches, float *first_bad, bool single_correction, hipStream_t s)
{
    if (single_correction)
        hipLaunchKernelGGL(divide_selftest_kernel<true>, dim3(blocks), dim3(kBlock), 0, s, seed, per_thread, d_lo, d_hi, n_lo, n_hi,
                           mismatches, first_bad);
    else
        hipLaunchKernelGGL(divide_selftest_kernel<false>, dim3(blocks), dim3(kBlock), 0, s, seed, per_thread, d_lo, d_hi, n_lo, n_hi,
                           mismatches, first_bad);
    return hipGetLastError();
}

// stride-3 host layout <-> 16-byte device records
__global__ __launch_bounds__(kBlock) void pack_kernel(uint32_t count, const float *__restrict__ xyz, float4 *__restrict__ rec)
{
    const uint32_t l = blockIdx.x * (uint32_t)kBlock + threadIdx.x;
    if (l >= count) return;
    rec[l] = make_float4(xyz[3 * (size_t)l], xyz[3 * (size_t)l + 1], xyz[3 * (size_t)l + 2], 0.f);
}

__global__ __launch_bounds__(kBlock) void unpack_kernel(uint32_t count, const float4 *__restrict__ rec, float *__restrict__ xyz)
{
    const uint32_t l = blockIdx.x * (uint32_t)kBlock + threadIdx.x;
    if (l >= count) return;
    const float4 r = rec[l];
    xyz[3 * (size_t)l] = r.x;
    xyz[3 * (size_t)l + 1] = r.y;
    xyz[3 * (size_t)l + 2] = r.z;
}

#endif  // !NBK_NOSLP_TU

// ------------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------------
static inline uint32_t ceil_div(uint32_t a, uint32_t b) { return (a + b - 1u) / b; }

template <int TJ, int U, int S>
static hipError_t launch_strict_t(const StepArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL((step_strict_kernel<TJ, U, S>), dim3(ceil_div(a.count, kBlock / S)), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

#define NBK_CASE(T, U_, S_) \
    if (tile == T && unroll == U_ && lanes == S_) return launch_strict_t<T, U_, S_>(a, s)

#ifdef NBK_NOSLP_TU
// This translation unit is the same source compiled with -DNBK_NOSLP_TU -fno-slp-vectorize: it holds only the
// j-parallel STRICT shapes (S > 1).  Without SLP the x/y running-sum adds stay scalar v_add_f32, so the row_shl DPP
// fetch folds into them (v_add_f32_dpp); packed v_pk_add_f32 cannot take a DPP operand and would need a mov per term.
hipError_t launch_strict_jp(const StepArgs &a, uint32_t tile, uint32_t unroll, uint32_t lanes, hipStream_t s)
{
    // unroll counts groups of S pairs in flight
    NBK_CASE(256, 4, 2);
    NBK_CASE(256, 2, 2);
    NBK_CASE(256, 4, 4);
    NBK_CASE(256, 2, 4);
    NBK_CASE(256, 4, 8);
    NBK_CASE(256, 2, 8);
    NBK_CASE(256, 4, 16);
    NBK_CASE(256, 2, 16);
    NBK_CASE(1024, 4, 2);
    NBK_CASE(1024, 4, 4);
    NBK_CASE(1024, 4, 8);
    NBK_CASE(1024, 2, 8);
    NBK_CASE(1024, 2, 16);
    return hipErrorInvalidValue;
}
#else
hipError_t launch_strict_pc(const StepArgs &a, uint32_t producers, hipStream_t s)
{
    if (producers == 8)
        hipLaunchKernelGGL(step_strict_pc_kernel<8>, dim3(ceil_div(a.count, 64)), dim3(PcShape<8>::kThreads), 0, s, a);
    else if (producers == 14)
        hipLaunchKernelGGL(step_strict_pc_kernel<14>, dim3(ceil_div(a.count, 64)), dim3(PcShape<14>::kThreads), 0, s, a);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_strict(const StepArgs &a, uint32_t tile, uint32_t unroll, uint32_t lanes, hipStream_t s)
{
    if (lanes > 1) return launch_strict_jp(a, tile, unroll, lanes, s);
    NBK_CASE(256, 4, 1);
    NBK_CASE(256, 8, 1);
    NBK_CASE(512, 4, 1);
    NBK_CASE(512, 8, 1);
    NBK_CASE(1024, 4, 1);
    NBK_CASE(1024, 8, 1);
    NBK_CASE(1024, 16, 1);
    return hipErrorInvalidValue;
}
#endif
#undef NBK_CASE

#ifndef NBK_NOSLP_TU
template <int TJ, int IB>
static hipError_t launch_fast_t(const StepArgs &a, uint32_t slices, hipStream_t s)
{
    hipLaunchKernelGGL((step_fast_kernel<TJ, IB>), dim3(ceil_div(a.count, kBlock * IB), slices), dim3(kBlock), 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || slices == 1) return e;
    hipLaunchKernelGGL(integrate_partials_kernel, dim3(ceil_div(a.count, kBlock)), dim3(kBlock), 0, s, a, slices);
    return hipGetLastError();
}

hipError_t launch_fast(const StepArgs &a, uint32_t tile, uint32_t ib, uint32_t slices, hipStream_t s)
{
#define NBK_CASE(T, I) \
    if (tile == T && ib == I) return launch_fast_t<T, I>(a, slices, s)
    NBK_CASE(256, 1);
    NBK_CASE(256, 2);
    NBK_CASE(256, 4);
    NBK_CASE(512, 1);
    NBK_CASE(512, 2);
    NBK_CASE(512, 4);
    NBK_CASE(1024, 1);
    NBK_CASE(1024, 2);
    NBK_CASE(1024, 4);
#undef NBK_CASE
    return hipErrorInvalidValue;
}

hipError_t launch_instances(uint32_t count, const float4 *pos, const float4 *vel, float4 *inst, hipStream_t s)
{
    hipLaunchKernelGGL(instances_kernel, dim3(ceil_div(count, kBlock)), dim3(kBlock), 0, s, count, pos, vel, inst);
    return hipGetLastError();
}

hipError_t launch_pack(uint32_t count, const float *xyz, float4 *rec, hipStream_t s)
{
    hipLaunchKernelGGL(pack_kernel, dim3(ceil_div(count, kBlock)), dim3(kBlock), 0, s, count, xyz, rec);
    return hipGetLastError();
}

hipError_t launch_unpack(uint32_t count, const float4 *rec, float *xyz, hipStream_t s)
{
    hipLaunchKernelGGL(unpack_kernel, dim3(ceil_div(count, kBlock)), dim3(kBlock), 0, s, count, rec, xyz);
    return hipGetLastError();
}

#endif  // !NBK_NOSLP_TU

}  // namespace nbk
