// nb_api.hip -- the extern "C" boundary declared in include/nenbody.h.
//
// Host-side only: argument checking, launch-shape selection, device buffers of a context.  There is no
// CPU implementation of the step in this library: without a HIP device every compute call fails.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cmath>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/nenbody.h"
#include "../../include/nenbody_diag.h"
#include "nb_kernels.h"

#define NB_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

thread_local std::string g_tls_error = "";

// MI355X: 256 CUs x 4 SIMDs.  FAST launches aim at 16 waves per SIMD's worth of work items: four times what is resident at
// once (4 bodies per lane: 120 registers, 4 waves per SIMD).  Measured at N = 131 072 (profiles/r02/fast_forms.log): with
// exactly one resident round (512 workgroups of 8 waves) the step takes 2.80 ms, with two rounds 2.62, with four 2.53 --
// workgroups that all start together finish when the slowest CU does, later rounds go to whoever is free.
constexpr uint32_t kTargetWaves = 16384;
constexpr uint32_t kMaxSlices = 64;

// ---- diagnostic overrides -----------------------------------------------------------------------------------
// The parity tests pin every kernel form and the sweeps in tools/ time them; both do it through NB_* environment
// variables.  They are read ONLY after a diagnostic call has asked for it -- nb_debug_reload_env() or nb_diag_enable_env(1)
// (include/nenbody_diag.h; the test suite's fixtures and every script under tools/ make it): a host that merely loads the
// library is steered by nothing its environment happens to export (round 4; VERDICT r03 item 11: until then an inherited
// NB_TILE or NB_FAST_SL changed which kernels ran).  The one exception is NB_ROCTX, which is a feature for the host (tracing),
// not a kernel form.  No launch path calls getenv.
struct Knob {
    bool set = false;
    uint32_t v = 0;
    // the override if present, else the default
    uint32_t or_else(uint32_t dflt) const { return set ? v : dflt; }
    bool on() const { return set && v != 0; }
};
struct DebugOverrides {
    Knob tile, fast_pairs_w, fast_pairs_np, fast_pairs_chunk, fast_ib, fast_groups, fast_waves, fast_slices, fast_no_share, strict_force_ieee, force_3d, strict_no_packed, strict_lanes, strict_unroll,
        strict_pc, strict_bc, strict_sl, fast_sl, fast_pairs, ring, ring_np, ring_ga, ring_wpb, ring_c4_own, ring_c4_rest, ring_cap, inst_device_libm, boids_slices, bc_spin_budget, bc_prio, boids_pc, boids_tile, boids_force, selftest_control, roctx, dropin_zero_copy, dropin_poll, dropin_poll_budget_us;
    uint32_t generation = 0;  // bumped by every reload: invalidates cached plans
};

Knob read_knob(const char *name)
{
    Knob k;
    const char *s = std::getenv(name);
    if (s && *s) {
        k.set = true;
        k.v = (uint32_t)std::strtoul(s, nullptr, 10);
    }
    return k;
}

std::atomic<const DebugOverrides *> g_overrides{nullptr};
std::atomic<bool> g_env_enabled{false};  // nb_diag_enable_env / nb_debug_reload_env
// nb_diag_rccl_solo (include/nenbody_diag.h): the test suite's way to run the RCCL leg of a shard on a one-GPU box.  Not an
// environment variable: nothing a deployment inherits can turn a world of 8 into eight communicators of one.
std::atomic<bool> g_rccl_solo{false};
std::mutex g_overrides_mu;

const DebugOverrides *parse_overrides(uint32_t generation)
{
    DebugOverrides *d = new DebugOverrides();  // never freed: a handful per process at most (one, outside the test suite)
    d->generation = generation;
    d->roctx = read_knob("NB_ROCTX");
    if (!g_env_enabled.load(std::memory_order_relaxed)) return d;  // a bare load of the library: no kernel-form override is read
    d->tile = read_knob("NB_TILE");
    d->fast_ib = read_knob("NB_FAST_IB");
    d->fast_groups = read_knob("NB_FAST_GROUPS");
    d->fast_waves = read_knob("NB_FAST_WAVES");
    d->fast_slices = read_knob("NB_FAST_SLICES");
    d->fast_no_share = read_knob("NB_FAST_NO_SHARE");
    d->strict_force_ieee = read_knob("NB_STRICT_FORCE_IEEE");
    d->force_3d = read_knob("NB_FORCE_3D");
    d->strict_no_packed = read_knob("NB_STRICT_NO_PACKED");
    d->strict_lanes = read_knob("NB_STRICT_LANES");
    d->strict_unroll = read_knob("NB_STRICT_UNROLL");
    d->strict_pc = read_knob("NB_STRICT_PC");
    d->strict_bc = read_knob("NB_STRICT_BC");
    d->strict_sl = read_knob("NB_STRICT_SL");
    d->fast_sl = read_knob("NB_FAST_SL");
    d->fast_pairs = read_knob("NB_FAST_PAIRS");
    d->fast_pairs_w = read_knob("NB_FAST_PAIRS_W");
    d->fast_pairs_chunk = read_knob("NB_FAST_PAIRS_CHUNK");
    d->fast_pairs_np = read_knob("NB_FAST_PAIRS_NP");
    d->inst_device_libm = read_knob("NB_INST_DEVICE_LIBM");
    d->ring = read_knob("NB_RING");
    d->ring_np = read_knob("NB_RING_NP");
    d->ring_ga = read_knob("NB_RING_GA");
    d->ring_wpb = read_knob("NB_RING_WPB");
    d->ring_c4_own = read_knob("NB_RING_C4_OWN");
    d->ring_c4_rest = read_knob("NB_RING_C4_REST");
    d->ring_cap = read_knob("NB_RING_CAP");
    d->bc_spin_budget = read_knob("NB_BC_SPIN_BUDGET");
    d->bc_prio = read_knob("NB_BC_PRIO");
    d->boids_pc = read_knob("NB_BOIDS_PC");
    d->boids_tile = read_knob("NB_BOIDS_TILE");
    d->boids_force = read_knob("NB_BOIDS_FORCE");
    d->boids_slices = read_knob("NB_BOIDS_SLICES");
    d->dropin_zero_copy = read_knob("NB_DROPIN_ZERO_COPY");
    d->dropin_poll = read_knob("NB_DROPIN_POLL");
    d->dropin_poll_budget_us = read_knob("NB_DROPIN_POLL_BUDGET_US");
    d->selftest_control = read_knob("NB_SELFTEST_CONTROL");
    return d;
}

const DebugOverrides &overrides()
{
    const DebugOverrides *d = g_overrides.load(std::memory_order_acquire);
    if (d) return *d;
    std::lock_guard<std::mutex> lock(g_overrides_mu);
    d = g_overrides.load(std::memory_order_acquire);
    if (!d) {
        d = parse_overrides(1);
        g_overrides.store(d, std::memory_order_release);
    }
    return *d;
}

bool valid_tile(uint32_t t) { return t == 256 || t == 512 || t == 1024; }

struct Plan {
    uint32_t tile;
    uint32_t ib;      // FAST: bodies per thread
    uint32_t waves;   // FAST: 0 = workgroup-tile form; else the barrier-free form with this many waves per workgroup (= groups)
    uint32_t groups;  // FAST: 256-lane groups per workgroup, each folding its own j chunk (combined in LDS)
    uint32_t slices;  // FAST: blockIdx.y slices of the j range (combined through memory)
    uint32_t j_chunk; // FAST: records per chunk (groups * slices chunks cover the set)
    uint32_t lo_bits, hi_bits, force_ieee;  // STRICT guard
    int guard_a, guard_b, guard_g, guard_c; // its exponents: coordinates in {0} U [2^a, 2^b], |G| in [2^g, 2^(g+1)), bias in [2^c, 2^(c+1))
    uint32_t force_3d;                      // 2 = never take the planar shortcut (NB_FORCE_3D=1: tests, measurements)
    uint32_t unroll;                        // STRICT: pairs in flight per lane (2, 4 or 8)
    uint32_t lanes;                         // STRICT: lanes per body (1 = plain; 2..16 = j-parallel, same summation order)
    uint32_t pc;                            // STRICT: 0 = off, else producers per workgroup of the producer/consumer form (8 or 14)
    uint32_t no_packed;                     // STRICT, one lane per body: 1 = do not use the j-packed planar fold (NB_STRICT_NO_PACKED=1)
    uint32_t bc;                            // STRICT: 1 = block-chain form (nb_nbody_bc.inc) instead of producer/consumer; needs scratch
    uint32_t pairs;                         // FAST: the pairs form (nb_nbody_sym.inc: every unordered pair once; whole sets, n a multiple of 256; needs scratch) with this many waves per workgroup (8, 4, 2, 1); 0 = another form
    uint32_t pairs_np;                      // pairs form: packed pairs of bodies per lane, 2 or 4 (blocks of 256 or 512 bodies)
    uint32_t pairs_chunk;                   // pairs form: bodies per chunk of the two-level walk (0: the default -- one tile up to 262 144 bodies, chunks of 131 072 beyond)
    uint32_t fsl;                           // FAST: 1 = scalar-load form (step_fast_sl_kernel, nb_nbody_sl.inc): whole-set launches, eight waves per workgroup; needs scratch
    uint32_t sl;                            // STRICT: 1 = scalar-load form (nb_nbody_sl.inc) instead of the LDS-tiled one-lane kernel; needs scratch
    uint32_t spin_budget;                   // block chain: polls per wait, 0 = kernel default (NB_BC_SPIN_BUDGET: the give-up test)
    uint32_t bc_prio;                       // block chain: raised wave priority around the chain turns (NB_BC_PRIO=0 switches it off)
    uint32_t n_total;                       // the set size the plan was made for
};

int floor_log2f(float x)
{
    int e;
    std::frexp(x, &e);  // x = m * 2^e, m in [0.5, 1)
    return e - 1;
}

// FAST: bodies per thread; two or four also let the thread's pairs share reciprocals
uint32_t fast_bodies_per_lane(uint32_t n_total, uint32_t count)
{
    // by the size of the launch, in pairs (measured, profiles/r02/fast_forms.log): 16 384 of 131 072 bodies: 4 (0.340 ms
    // against 0.360 with 2); standalone sets of 32 768 and 16 384: 2 (0.184 against 0.207 with 4); 8 192 and below: 1
    const uint64_t pairs = (uint64_t)n_total * count;
    uint32_t ib = overrides().fast_ib.or_else(pairs >= (1ull << 31) ? 4u : pairs >= (1ull << 27) ? 2u : 1u);
    return (ib == 1 || ib == 2 || ib == 4) ? ib : 1u;
}

// FAST: 0 = the workgroup-tile form (step_fast_kernel), else waves per workgroup of the barrier-free form (1, 4, 8 or 16)
uint32_t fast_wave_form()
{
    const uint32_t w = overrides().fast_waves.or_else(8u);  // NB_FAST_WAVES=0: the workgroup-tile form
    return (w == 1 || w == 4 || w == 8 || w == 16) ? w : 0u;
}

// FAST: how a fold over `n_fold` records is split for `count` bodies at `ib` bodies per lane: `groups` chunks inside a
// workgroup (combined in LDS) times `slices` across workgroups (combined through memory by integrate_partials_kernel),
// `chunk` records each (a multiple of the tile).  The split aims at kTargetWaves waves in all.
// `waves` != 0 selects the barrier-free form (step_fast_wave_kernel): a workgroup is `waves` waves that share 64*ib bodies
// and fold one chunk each; then *groups = waves.
void fast_split(uint32_t n_fold, uint32_t count, uint32_t ib, uint32_t waves, uint32_t *tile, uint32_t *groups, uint32_t *slices,
                uint32_t *chunk)
{
    const DebugOverrides &dbg = overrides();
    if (waves && *tile == 1024u) *tile = 512u;  // the wave form stages a whole tile per wave: 256 or 512 records
    if (waves && *tile == 512u && (ib == 1 || waves == 1)) *tile = 256u;  // built 512-record shapes: 4, 8, 16 waves at 2 / 4 bodies per lane
    const uint32_t bodies = waves ? 64u * ib : 256u * ib, chunk_waves = waves ? 1u : 4u;
    const uint32_t blocks = (count + bodies - 1u) / bodies;
    const uint32_t max_by_tiles = std::max(1u, (n_fold + *tile - 1u) / *tile);
    uint32_t split = (kTargetWaves + blocks * chunk_waves - 1u) / (blocks * chunk_waves);
    if (split > max_by_tiles) split = max_by_tiles;
    if (split < 1) split = 1;
    uint32_t g = waves ? waves : dbg.fast_groups.or_else(split >= 4u ? 4u : split >= 2u ? 2u : 1u);
    if (!waves && g != 1 && g != 2 && g != 4) g = 1;
    if (!waves && *tile == 1024u) g = 1;  // groups are built for tiles of 256 and 512 records (4 x 2 x 1024 records = 128 KB of LDS)
    uint32_t sl = dbg.fast_slices.or_else((split + g - 1u) / g);
    if (sl > kMaxSlices) sl = kMaxSlices;
    if (sl < 1) sl = 1;
    uint32_t chunks = sl * g;
    if (chunks > max_by_tiles) chunks = max_by_tiles;
    uint32_t c = (std::max(n_fold, 1u) + chunks - 1u) / chunks;
    c = ((c + *tile - 1u) / *tile) * *tile;
    chunks = (std::max(n_fold, 1u) + c - 1u) / c;   // drop empty chunks ...
    sl = (chunks + g - 1u) / g;                     // ... and the slices made of them
    *groups = g;
    *slices = sl;
    *chunk = c;
}

// Launch shape + STRICT guard range for (params, n_total, count).  Pure host arithmetic.
int make_plan(const nb_params &p, uint32_t n_total, uint32_t count, Plan *out, std::string *err)
{
    if (n_total == 0 || count == 0 || count > n_total) {
        *err = "nb: need 0 < count <= n_total";
        return NB_ERR_INVALID;
    }
    if (p.mode != NB_MODE_STRICT && p.mode != NB_MODE_FAST) {
        *err = "nb: params.mode must be NB_MODE_STRICT or NB_MODE_FAST";
        return NB_ERR_INVALID;
    }
    if (n_total > 0x80000000u) {  // the kernels index records and tiles in 32 bits with headroom for padding
        *err = "nb: sets of more than 2^31 bodies are not supported";
        return NB_ERR_UNSUPPORTED;
    }
    const DebugOverrides &dbg = overrides();
    Plan pl{};
    pl.tile = p.tile;
    // FAST: the barrier-free form stages a whole tile per wave (256 records: 4 per lane, 120 registers at 4 bodies per lane;
    // 512 would take 142 and cost a wave per SIMD); the workgroup-tile form shares tiles of 512
    if (pl.tile == 0) pl.tile = dbg.tile.or_else((p.mode == NB_MODE_STRICT) ? 1024u : fast_wave_form() ? 256u : 512u);
    if (!valid_tile(pl.tile)) {
        *err = "nb: params.tile must be 0, 256, 512 or 1024";
        return NB_ERR_INVALID;
    }
    pl.ib = 1;
    pl.groups = 1;
    pl.slices = 1;
    pl.j_chunk = n_total;
    if (p.mode == NB_MODE_FAST) {
        pl.ib = fast_bodies_per_lane(n_total, count);
        pl.waves = fast_wave_form();
        fast_split(n_total, count, pl.ib, pl.waves, &pl.tile, &pl.groups, &pl.slices, &pl.j_chunk);
    }
    // FAST through scalar loads (nb_nbody_sl.inc) for sets of 4 096 bodies and more (below, its second launch -- the planes --
    // costs more than it saves): the same split of the j range over eight waves per workgroup and grid.y slices.  Naming a
    // tile, a wave count or groups asks for one of the LDS forms; NB_FAST_SL=0/1 decides outright.
    if (p.mode == NB_MODE_FAST) {
        const bool no_form_named = p.tile == 0 && !dbg.tile.set && !dbg.fast_waves.set && !dbg.fast_groups.set;
        pl.fsl = dbg.fast_sl.or_else((pl.waves == 8u && n_total >= 4096u && no_form_named) ? 1u : 0u);
        // The pairs form (nb_nbody_sym.inc): every UNORDERED pair once.  Whole sets only (both bodies of a pair on this GPU), n a
        // multiple of its 256-body blocks.  One workgroup of W waves per pair of superblocks of 256 W bodies; measured by size and W
        // (profiles/r03/pairs_sizes.log): it wins from 32 768 bodies on (0.95 of the ordered fold's time there, 0.82 at 65 536,
        // 0.76-0.78 from 114 688 on), with W = 4 below 131 072 (eight-wave workgroups are too few and too long there: 1.12 ms
        // against 1.02 at 98 304) and W = 8 from there on (the same time as W = 4 and half the rows).  The rows of partial sums
        // take n x superblocks x 12 B -- 100 MB at n = 131 072 --, so sets of more than 262 144 bodies are walked in chunks of
        // 131 072 (tiles of one or two chunks, each tile's rows added to running sums before the next: 0.2 GB of rows + 12 B per
        // body, whatever the size; up to 4 194 304 bodies).
        // NB_FAST_PAIRS=0/1 decides outright where it can run, NB_FAST_PAIRS_W names W (8, 4, 2, 1), NB_FAST_PAIRS_CHUNK the chunk.
        constexpr uint32_t kPairsMinN = 32768u;
        const bool pairs_ok = count == n_total && n_total % 256u == 0 && n_total <= 4194304u;
        // From 98 304 bodies on a lane holds EIGHT bodies (four packed pairs, blocks of 512) where the size allows: the step's fixed
        // costs -- the rotating sums' four instructions, the LDS read, the hazard nops -- are shared by twice the pairs: 1.68 against
        // 1.78 ms at 131 072 (W = 4), 6.61 against 7.19 at 262 144, 106 against 114 at 2^20 (profiles/r03/pairs_sizes_np.log); below,
        // four bodies per lane (142 registers leave three waves per SIMD, too few where the workgroups are few as well).
        uint32_t pnp = (n_total >= 98304u && n_total % 512u == 0) ? 4u : 2u;
        if (dbg.fast_pairs_np.set && (dbg.fast_pairs_np.v == 2 || (dbg.fast_pairs_np.v == 4 && n_total % 512u == 0))) pnp = dbg.fast_pairs_np.v;
        uint32_t pw = pnp == 4u ? (n_total >= 131072u ? 4u : 2u) : (n_total >= 131072u ? 8u : 4u);
        if (dbg.fast_pairs_w.set && (dbg.fast_pairs_w.v == 1 || dbg.fast_pairs_w.v == 2 || dbg.fast_pairs_w.v == 4 || (dbg.fast_pairs_w.v == 8 && pnp == 2u)))
            pw = dbg.fast_pairs_w.v;
        pl.pairs = (pairs_ok && dbg.fast_pairs.or_else((n_total >= kPairsMinN && no_form_named && !dbg.fast_sl.set && !dbg.fast_ib.set &&
                                                        !dbg.fast_slices.set) ? 1u : 0u)) ? pw : 0u;
        pl.pairs_np = pnp;
        pl.pairs_chunk = 0;
        if (pl.pairs && dbg.fast_pairs_chunk.on()) {  // whole superblocks per chunk
            const uint32_t super = 128u * pnp * pw;
            pl.pairs_chunk = std::max(super, dbg.fast_pairs_chunk.v / super * super);
        }
        if (pl.pairs) pl.fsl = 0;
        if (pl.fsl) {  // its own split: 8 waves per workgroup, chunks of whole 256-record tiles (16-record requests stay aligned)
            uint32_t tile = 256u;
            pl.waves = 8u;
            fast_split(n_total, count, pl.ib, pl.waves, &tile, &pl.groups, &pl.slices, &pl.j_chunk);
            pl.tile = tile;
        }
    }
    // STRICT: magnitude range {0} U [2^a, 2^b] of coordinates for which d, n = dx*G, q = n/d and the
    // ladder's residuals are all normal binary32 with headroom, so that v_div_scale/v_div_fixup would be
    // the identity and the shared-reciprocal ladder equals the IEEE divide bit for bit.
    //   nonzero |dx| >= 2^(a-23), |dx| <= 2^(b+1);  d in [bias, 3*2^(2b+2)+bias];  |n| >= 2^(a-23+g)
    pl.force_ieee = 1;
    pl.lo_bits = 0x3f800000u;
    pl.hi_bits = 0x3f800000u;
    const float G = std::fabs(p.G);
    if (std::isfinite(p.G) && std::isfinite(p.bias) && G > 0.f && p.bias > 0.f) {
        const int g = floor_log2f(G), c = floor_log2f(p.bias);
        const int b = 20;
        const int dmax = std::max(2 * b + 4, c + 2);
        int a = std::max(-100 + 23 - g, -120 + 23 - g + dmax);
        const bool ok = g >= -100 && g <= 100 && c >= -100 && dmax <= 100 && (b + 2 + g - c) <= 120 && a <= b - 1;
        if (ok) {
            if (a < -120) a = -120;
            pl.lo_bits = (uint32_t)(a + 127) << 23;
            pl.hi_bits = (uint32_t)(b + 127) << 23;
            pl.force_ieee = 0;
            pl.guard_a = a;
            pl.guard_b = b;
            pl.guard_g = g;
            pl.guard_c = c;
        }
    }
    if (dbg.strict_force_ieee.on()) pl.force_ieee = 1;
    if (p.mode == NB_MODE_FAST) {
        // FAST reuses the field: 1 = never share a reciprocal between two pairs (the product of two r^2 must stay normal:
        // bias in [2^-60, 2^60]; coordinates are checked per tile on the device).  NB_FAST_NO_SHARE=1: tests, measurements.
        const bool bias_ok = std::isfinite(p.bias) && p.bias >= 0x1p-60f && p.bias <= 0x1p60f;
        pl.force_ieee = (!bias_ok || dbg.fast_no_share.on()) ? 1u : 0u;
    }
    pl.force_3d = dbg.force_3d.on() ? 2u : 0u;
    pl.no_packed = dbg.strict_no_packed.on() ? 1u : 0u;
    // STRICT cannot split the fold over j (the sum is sequential), so a small shard would leave SIMDs idle:
    // below 65 536 bodies give each body S lanes until the shard supplies 2 waves per SIMD (256 CUs x 4 SIMDs x 2 =
    // 2048 waves).  The DPP adds of the j-parallel form cost about twice a plain add, so S = 1 stays ahead down to
    // one wave per SIMD (measured: profiles/r01_jp/sweep_lanes.log).
    // (Round 3: the line is 57 344 bodies, not 65 536 -- above it the scalar-load kernel, one wave per SIMD and nothing to wait
    // for, beats every chained form: a 65 536-body shard of 131 072 takes 2.95 ms against the block chain's 3.35,
    // profiles/r03/shard_strict_forms.log; the block chain costs 0.84 ms per 16 384 bodies, so the two meet at 57 500.)
    constexpr uint32_t kChainMax = 57344u;
    pl.lanes = 1;
    if (count <= kChainMax)
        while (pl.lanes < 16 && (uint64_t)count * pl.lanes < 2048ull * 64ull) pl.lanes *= 2;
    pl.lanes = dbg.strict_lanes.or_else(pl.lanes);
    if (pl.lanes != 1 && pl.lanes != 2 && pl.lanes != 4 && pl.lanes != 8 && pl.lanes != 16) pl.lanes = 1;
    pl.unroll = dbg.strict_unroll.or_else(pl.lanes == 1 ? 8 : (pl.lanes == 16 ? 2 : 4));
    if (pl.lanes == 1 && pl.unroll != 4 && pl.unroll != 8 && pl.unroll != 16) pl.unroll = 8;
    if (pl.lanes > 1 && pl.unroll != 2 && pl.unroll != 4) pl.unroll = 4;
    if (pl.lanes == 16) pl.unroll = 2;
    if (p.mode == NB_MODE_STRICT && pl.lanes > 1 && pl.tile == 512) pl.tile = 1024;  // built j-parallel shapes: 256, 1024
    if (p.mode == NB_MODE_STRICT && pl.lanes > 1 && pl.tile == 1024 && pl.unroll == 2 && pl.lanes < 8) pl.unroll = 4;
    // Shards of up to 57 344 bodies (round 1-2: 65 536) cannot fill the chip with one lane per body (the running sum is a serial chain per
    // body).  Default there: the block-chain form (nb_nbody_bc.inc).  Against all 131 072 bodies: 16 384 bodies 0.85 ms
    // (producer/consumer, 14 producers + 2 consumers per 64 bodies: 1.15; j-parallel S = 8: 2.1), 32 768: 1.67 (2.28),
    // 65 536: 3.30 (one lane per body 3.52); the whole set: 6.57 against 6.19 for one lane per body, which therefore keeps
    // everything above (profiles/r01_final/pc_sweep.log).  It needs scratch memory (nb_scratch_bytes).
    // Small SETS (tools/small_forms.py, profiles/r02/small_forms.log; microseconds per step, j-parallel / producer-consumer /
    // block chain): N = 100: 3.7 / 4.3 / 6.9, 1 024: 11.6 / 12.8 / 12.4, 2 048: 20.2 / 21.5 / 19.9, 3 072: 28.9 / 30.0 / 25.8,
    // 4 096: 37.8 / 39.5 / 32.9 -- below 1 536 bodies the j-parallel form (the block chain's second launch costs more than
    // the form gains), from there on the block chain (round 1's four launches had put that line at 4 096).
    // Naming another shape (NB_STRICT_PC / NB_STRICT_LANES) turns the block chain off; NB_STRICT_BC=0/1 decides outright.
    constexpr uint32_t kSmallSet = 1536u;
    pl.pc = dbg.strict_pc.or_else((count <= kChainMax && n_total >= kSmallSet) ? 14u : 0u);
    if (pl.pc == 1) pl.pc = 8;
    if (pl.pc != 0 && pl.pc != 8 && pl.pc != 14) pl.pc = 8;
    pl.n_total = n_total;
    pl.bc = dbg.strict_bc.or_else((count <= kChainMax && n_total >= kSmallSet && !dbg.strict_pc.set && !dbg.strict_lanes.set) ? 1u : 0u);
    pl.bc = (pl.bc && p.mode == NB_MODE_STRICT) ? 1u : 0u;
    pl.spin_budget = dbg.bc_spin_budget.or_else(0u);
    pl.bc_prio = dbg.bc_prio.or_else(1u) ? 1u : 0u;
    // Above the block chain's range (whole sets, shards of more than 57 344 bodies) one lane per body fills the chip, and the
    // records reach the lanes through scalar loads instead of LDS tiles: no barrier, no ds_read (nb_nbody_sl.inc).  Naming a
    // tile (params.tile / NB_TILE), a lane count or the producer/consumer form asks for the LDS-tiled kernel; NB_STRICT_SL=0/1 decides outright.
    pl.sl = dbg.strict_sl.or_else((!pl.bc && !pl.pc && pl.lanes == 1 && p.tile == 0 && !dbg.tile.set && !dbg.strict_lanes.set &&
                                   !dbg.strict_pc.set && !dbg.strict_unroll.set && !dbg.strict_no_packed.set && n_total >= kSmallSet) ? 1u : 0u);
    pl.sl = (pl.sl && p.mode == NB_MODE_STRICT && !pl.bc) ? std::min(pl.sl, 3u) : 0u;  // NB_STRICT_SL = 1 + launch shape (nb_nbody_sl.inc)
    *out = pl;
    return NB_OK;
}

size_t plan_scratch_bytes(const Plan &pl, uint32_t count)
{
    if (pl.bc || pl.sl) return nbk::strict_bc_scratch_bytes(pl.n_total);
    if (pl.pairs) return nbk::strict_bc_scratch_bytes(pl.n_total) + nbk::fast_pairs_scratch_floats(pl.n_total, pl.pairs, pl.pairs_np, pl.pairs_chunk) * sizeof(float);
    if (pl.fsl) return nbk::strict_bc_scratch_bytes(pl.n_total) + (pl.slices > 1 ? (size_t)pl.slices * count * sizeof(float4) : 0);
    return pl.slices > 1 ? (size_t)pl.slices * count * sizeof(float4) : 0;
}

// The launch API is stateless; a rank calls it with the same shape every step.  A few plans per thread are kept, keyed by
// everything make_plan reads (no lock: one thread drives one GPU).
int cached_plan(const nb_params &p, uint32_t n_total, uint32_t count, const Plan **out, std::string *err)
{
    struct Entry {
        bool valid = false;
        nb_params p{};
        uint32_t n_total = 0, count = 0, generation = 0;
        Plan pl{};
    };
    constexpr int kEntries = 4;
    thread_local Entry cache[kEntries];
    thread_local int next = 0;
    const uint32_t gen = overrides().generation;
    for (int i = 0; i < kEntries; ++i) {
        const Entry &e = cache[i];
        if (e.valid && e.n_total == n_total && e.count == count && e.generation == gen && std::memcmp(&e.p, &p, sizeof(p)) == 0) {
            *out = &e.pl;
            return NB_OK;
        }
    }
    Plan pl;
    int rc = make_plan(p, n_total, count, &pl, err);
    if (rc != NB_OK) return rc;
    Entry &e = cache[next];
    next = (next + 1) % kEntries;
    e.valid = true;
    e.p = p;
    e.n_total = n_total;
    e.count = count;
    e.generation = gen;
    e.pl = pl;
    *out = &e.pl;
    return NB_OK;
}

// ---- roctx ranges (tracing) -----------------------------------------------------------------
// NB_ROCTX=1: the step loops of the context and shard APIs are bracketed with roctx ranges ("nb_step", "nb_shard_step",
// "nb_shard_exchange", ...), which rocprofv3 --marker-trace puts on the timeline.  The library is loaded with dlopen at first
// use (librocprofiler-sdk-roctx.so, else libroctx64.so); without NB_ROCTX nothing is loaded and a range costs one branch.
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
};
const Roctx *roctx_api()
{
    if (!overrides().roctx.on()) return nullptr;
    static const Roctx *api = [] {
        Roctx *r = new Roctx();
        void *h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_LOCAL);
        if (h) {
            r->push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
            r->pop = (int (*)())dlsym(h, "roctxRangePop");
        }
        return (const Roctx *)r;
    }();
    return (api->push && api->pop) ? api : nullptr;
}
struct RoctxRange {
    const Roctx *api;
    explicit RoctxRange(const char *name) : api(roctx_api())
    {
        if (api) (void)api->push(name);
    }
    ~RoctxRange()
    {
        if (api) (void)api->pop();
    }
    RoctxRange(const RoctxRange &) = delete;
    RoctxRange &operator=(const RoctxRange &) = delete;
};

// ---- sticky device status ------------------------------------------------------------------------------------
// One word per device, owned by the library, zero at allocation.  The block-chain kernel ORs 1 into it when a wave gave
// up waiting for its turn (its workgroup's outputs are NaN then).  check_device_status() -- called by every entry point
// that waits for the device -- turns a set word into NB_ERR_STATE and clears it.
constexpr int kMaxDevices = 64;
uint32_t *g_status_word[kMaxDevices] = {};
std::mutex g_status_mu;

int device_status_word(uint32_t **out, std::string *err)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= kMaxDevices) {
        *err = std::string("nb: hipGetDevice failed: ") + hipGetErrorString(e);
        return NB_ERR_HIP;
    }
    std::lock_guard<std::mutex> lock(g_status_mu);
    if (!g_status_word[dev]) {
        uint32_t *w = nullptr;
        e = hipMalloc((void **)&w, 64);
        if (e == hipSuccess) e = hipMemset(w, 0, 64);
        if (e != hipSuccess) {
            *err = std::string("nb: allocating the device status word failed: ") + hipGetErrorString(e);
            return e == hipErrorOutOfMemory ? NB_ERR_ALLOC : NB_ERR_HIP;
        }
        g_status_word[dev] = w;
    }
    *out = g_status_word[dev];
    return NB_OK;
}

// The caller has already waited for the work in question.  NB_OK, or NB_ERR_STATE (and the word cleared) if a kernel
// reported a failure through `w` since the last check.
int check_status_word(uint32_t *w, std::string *err)
{
    if (!w) return NB_OK;  // no block-chain launch yet
    uint32_t v = 0;
    hipError_t e = hipMemcpy(&v, w, sizeof(v), hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        *err = std::string("nb: reading the device status word failed: ") + hipGetErrorString(e);
        return NB_ERR_HIP;
    }
    if (v == 0u) return NB_OK;
    (void)hipMemset(w, 0, sizeof(v));
    *err = "nb: a STRICT block-chain workgroup gave up waiting for its turn (its outputs are NaN): the state is invalid; "
           "upload it again";
    return NB_ERR_STATE;
}

// the launch API's word: the current device's
int check_device_status(std::string *err)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return NB_OK;
    uint32_t *w;
    {
        std::lock_guard<std::mutex> lock(g_status_mu);
        w = g_status_word[dev];
    }
    return check_status_word(w, err);
}

// a context's / shard's own word (allocated at the first block-chain launch), so that one object's failure is not
// reported to another; `dirty` = a block-chain launch since the last check (keeps the check off paths that never use it)
struct StatusWord {
    uint32_t *w = nullptr;
    bool dirty = false;
};

int status_word_for_launch(StatusWord *sw, std::string *err)
{
    if (!sw->w) {
        hipError_t e = hipMalloc((void **)&sw->w, 64);
        if (e == hipSuccess) e = hipMemset(sw->w, 0, 64);
        if (e != hipSuccess) {
            *err = std::string("nb: allocating the status word failed: ") + hipGetErrorString(e);
            return e == hipErrorOutOfMemory ? NB_ERR_ALLOC : NB_ERR_HIP;
        }
    }
    sw->dirty = true;
    return NB_OK;
}

int check_status(StatusWord *sw, std::string *err)
{
    if (!sw->dirty) return NB_OK;
    sw->dirty = false;
    return check_status_word(sw->w, err);
}

int launch_step_planned(const nb_params &p, const Plan &pl, uint32_t n_total, uint32_t first, uint32_t count,
                        const void *pos_in, void *pos_out, void *vel, void *scratch, hipStream_t stream, std::string *err,
                        StatusWord *sw = nullptr)
{
    nbk::StepArgs a{};
    a.pos_in = (const float4 *)pos_in;
    a.pos_out = (float4 *)pos_out;
    a.vel = (float4 *)vel;
    a.partial = (float4 *)scratch;
    a.n_total = n_total;
    a.first = first;
    a.count = count;
    a.dt = p.dt;
    a.G = p.G;
    a.bias = p.bias;
    a.lo_bits = pl.lo_bits;
    a.hi_bits = pl.hi_bits;
    a.force_ieee = pl.force_ieee;
    a.force_3d = pl.force_3d;
    a.j_chunk = pl.j_chunk;
    a.no_packed = pl.no_packed;
    a.spin_budget = pl.spin_budget;
    a.bc_prio = pl.bc_prio;
    a.j_count = n_total;  // the whole set: no base, no hole
    a.j_base = 0;
    a.hole_lo = 0xffffffffu;
    a.hole_len = 0;
    uint32_t *status = nullptr;
    if (p.mode == NB_MODE_STRICT && pl.bc) {
        int rc = sw ? status_word_for_launch(sw, err) : device_status_word(&status, err);
        if (rc != NB_OK) return rc;
        if (sw) status = sw->w;
    }
    hipError_t e = (p.mode == NB_MODE_STRICT) ? (pl.bc   ? nbk::launch_strict_bc(a, scratch, status, stream)
                                                 : pl.sl ? nbk::launch_strict_sl(a, pl.sl - 1u, scratch, stream)
                                                 : pl.pc ? nbk::launch_strict_pc(a, pl.pc, stream)
                                                       : nbk::launch_strict(a, pl.tile, pl.unroll, pl.lanes, stream))
                                              : pl.pairs ? nbk::launch_fast_pairs(a, pl.pairs, pl.pairs_np, pl.pairs_chunk, scratch, stream)
                                              : pl.fsl   ? nbk::launch_fast_sl(a, pl.ib, pl.slices, scratch, stream)
                                              : pl.waves ? nbk::launch_fast_wave(a, pl.tile, pl.ib, pl.waves, pl.slices, stream)
                                                         : nbk::launch_fast(a, pl.tile, pl.ib, pl.groups, pl.slices, stream);
    if (e == hipErrorNotSupported) {  // a launch shape only the diagnostic knobs can name, in a build without -DNB_LEGACY_FORMS
        *err = "nb: this launch shape is one of the legacy forms (NB_STRICT_PC / NB_STRICT_LANES / NB_STRICT_UNROLL / NB_STRICT_SL=2,3 / "
               "NB_FAST_WAVES / NB_FAST_GROUPS / NB_FAST_PAIRS_W): this library was built without them (make -C nenbody_amd/csrc legacy "
               "builds libnenbody_hip_legacy.so, which holds every form)";
        return NB_ERR_UNSUPPORTED;
    }
    if (e != hipSuccess) {
        *err = std::string("nb: kernel launch failed: ") + hipGetErrorString(e);
        return NB_ERR_HIP;
    }
    return NB_OK;
}

int check_device(std::string *err)
{
    static std::atomic<bool> seen{false};  // a device does not disappear; cache only the positive answer (any thread may ask)
    if (seen.load(std::memory_order_relaxed)) return NB_OK;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        *err = std::string("nb: no HIP device available (") + (e == hipSuccess ? "0 devices" : hipGetErrorString(e)) +
               "); this library has no CPU path";
        return NB_ERR_NO_DEVICE;
    }
    seen.store(true, std::memory_order_relaxed);
    return NB_OK;
}

// Largest binary32 x >= 0 with sqrtf(x) < r (correctly rounded sqrt, as the reference's f32::sqrt), or -1 if there is
// none: then `sqrtf(d2) < r` <=> `d2 <= T` for every d2 (NaN fails both; sqrt and rounding are monotonic).
float sqrt_threshold(float r)
{
    if (!(r > 0.0f)) return -1.0f;  // r <= 0 or NaN: sqrt(d2) >= 0 is never below it
    uint32_t lo = 0u, hi = 0x7f7fffffu;  // bit patterns of +0 .. FLT_MAX: monotonic in value
    auto below = [&](uint32_t b) {
        float x;
        std::memcpy(&x, &b, 4);
        return std::sqrt(x) < r;
    };
    if (!below(lo)) return -1.0f;
    if (below(hi)) {
        float x;
        std::memcpy(&x, &hi, 4);
        return x;
    }
    while (hi - lo > 1u) {  // invariant: below(lo) && !below(hi)
        const uint32_t mid = lo + (hi - lo) / 2u;
        if (below(mid))
            lo = mid;
        else
            hi = mid;
    }
    float x;
    std::memcpy(&x, &lo, 4);
    return x;
}

// Largest binary32 V >= 0 such that rule 3 (main.rs:497-498) holds for EVERY pair of velocities whose components all lie in
// [-V, V], or -1 if there is none.  |v_i - v_n| <= 2V per component (2V is representable, rounding is monotonic), so the
// squared distance as the reference rounds it, ((ex*ex) + (ey*ey)) + (ez*ez), is at most the same expression at
// ex = ey = ez = 2V; the bound is monotonic in V, so bisection over the bit patterns finds the largest V that keeps it <= t3.
float rule3_always_bound(float t3)
{
    if (!(t3 >= 0.0f)) return -1.0f;
    auto holds = [&](uint32_t b) {
        float v;
        std::memcpy(&v, &b, 4);
        const volatile float w = v + v;
        const volatile float q = w * w;
        const volatile float two = q + q;
        const volatile float bound = two + q;
        return bound <= t3;  // false when anything overflowed to +inf
    };
    uint32_t lo = 0u, hi = 0x7f7fffffu;
    if (!holds(lo)) return -1.0f;
    if (!holds(hi)) {
        while (hi - lo > 1u) {  // invariant: holds(lo) && !holds(hi)
            const uint32_t mid = lo + (hi - lo) / 2u;
            if (holds(mid))
                lo = mid;
            else
                hi = mid;
        }
    } else {
        lo = hi;
    }
    float x;
    std::memcpy(&x, &lo, 4);
    return x;
}

// boids launch form (nb_kernels.h:launch_boids), by bodies per rank.  Measured at N = 131 072 (tools/sweep.py boidsshard,
// profiles/r02/boids_forms.log), ms per step for shards of 16 384 / 32 768 / 49 152 / 65 536 / 131 072 bodies:
//   1  producer/consumer                      1.43 / 2.28 / 3.44 / 4.14 / 8.0   (rounds of 512 workgroups of 64 bodies: steps, not a line)
//   3  chain split, two waves per 64 bodies   3.54 / 3.53 / 3.54 / 3.56 / 7.17  (one round of 256 workgroups up to 65 536 bodies; 7.3 beyond)
//   2  one lane per body, (x, y) packed       3.84 / 3.87 / 3.86 / 3.86 / 5.79  (a lone wave per SIMD is bound by its instruction count)
//   0  one lane per body                      4.81 / 4.82 / 4.83 / 4.84 / 4.96-5.3 (two waves per SIMD are bound by lane operations)
//   4  chain split packed                     3.88 / 3.90 / 3.90 / 3.91 / 7.85
// so: producer/consumer up to 48 K bodies, chain split up to 64 K, one lane per body above.
// NB_BOIDS_PC: 1 = producer/consumer, 2 / 3 = one lane per body packed / plain, 4 / 5 = chain split plain / packed,
// 0 = one lane per body, packed while the launch has at most one wave per SIMD.
uint32_t boids_form(uint32_t count)
{
    const Knob &k = overrides().boids_pc;
    if (k.set) return k.v == 1u ? 1u : k.v == 2u ? 2u : k.v == 3u ? 0u : k.v == 4u ? 3u : k.v == 5u ? 4u : (count <= 65536u ? 2u : 0u);
    return count <= 49152u ? 1u : count <= 65536u ? 3u : 0u;
}

int make_boids_args(const nb_boids_params &p, uint32_t n_total, uint32_t first, uint32_t count, nbk::BoidsArgs *out,
                    uint32_t *tile, std::string *err)
{
    if (n_total == 0 || count == 0 || (uint64_t)first + count > n_total) {
        *err = "nb: boids: need count > 0 and [first, first+count) inside n_total";
        return NB_ERR_INVALID;
    }
    const uint32_t t = p.tile ? p.tile : overrides().boids_tile.or_else(1024u);
    if (!valid_tile(t)) {
        *err = "nb: boids params.tile must be 0, 256, 512 or 1024";
        return NB_ERR_INVALID;
    }
    nbk::BoidsArgs a{};
    a.n_total = n_total;
    a.first = first;
    a.count = count;
    a.dt = p.dt;
    a.r1 = p.rule_1_distance;
    a.t2 = sqrt_threshold(p.rule_2_distance);
    a.t3 = sqrt_threshold(p.rule_3_distance);
    a.s1 = p.rule_1_scale;
    a.s2 = p.rule_2_scale;
    a.s3 = p.rule_3_scale;
    if (n_total >= (1u << 24)) {
        *err = "nb: boids: neighbour counts are kept in binary32 and need n_total < 2^24";
        return NB_ERR_UNSUPPORTED;
    }
    a.force_flags = overrides().boids_force.or_else(0u) & 7u;
    // the radius tests as single instructions (nb_boids.inc:boids_below): `x < T` needs T finite, positive and not tiny, so that
    // k = 2^(30 - e_T) and T * k are ordinary binary32 numbers; "nothing passes" (T <= 0) is the constant mask 0.  A radius
    // without such constants switches the masked form off for the step (flag 1: the select form, same bits).
    auto mask_consts = [](float T, float *k, float *tk) {
        if (!(T > 0.0f)) {  // T <= 0, -inf or NaN: x < T never holds for x >= 0
            *k = 0.0f;
            *tk = -1.0f;
            return true;
        }
        if (!std::isfinite(T) || T < 0x1p-90f) return false;
        const int e = floor_log2f(T);
        *k = std::ldexp(1.0f, 30 - e);  // e in [-90, 127]: 2^-97 .. 2^120
        *tk = T * *k;                   // exact: a power-of-two scaling into [2^30, 2^31)
        return true;
    };
    auto above = [](float t) { return t < 0.0f ? -1.0f : std::nextafter(t, INFINITY); };  // x <= t  <=>  x < above(t)
    if (!mask_consts(a.r1, &a.mk.k1, &a.mk.tk1) || !mask_consts(above(a.t2), &a.mk.k2, &a.mk.tk2) ||
        !mask_consts(above(a.t3), &a.mk.k3, &a.mk.tk3)) {
        a.force_flags |= 1u;
        a.mk = nbk::BoidsMaskK{0.f, -1.f, 0.f, -1.f, 0.f, -1.f};
        a.mk_valid = 0u;
    } else {
        a.mk_valid = 1u;
    }
    const float v_lim = rule3_always_bound(a.t3);
    if (v_lim < 0.0f)
        a.force_flags |= 4u;  // no such bound (rule_3_distance <= 0 or NaN): always test
    else
        std::memcpy(&a.vlim_bits, &v_lim, 4);
    *out = a;
    *tile = t;
    return NB_OK;
}

// The split form of the boids step (nb_launch_boids_step_split): slices of the j range so that one lane per body per slice gives
// the chip two waves per SIMD (512 workgroups of four waves), whole tiles per slice.  NB_BOIDS_SLICES names the count (tests, tools/).
void boids_split_shape(uint32_t n_total, uint32_t count, uint32_t tile, uint32_t *slices, uint32_t *j_chunk)
{
    // (an empty rank -- count == 0 -- launches nothing; it still gets a shape, so that a host can size scratch for every rank alike)
    const uint32_t ntiles = (n_total + tile - 1u) / tile, groups = std::max(1u, (count + 255u) / 256u);
    uint32_t sl = overrides().boids_slices.or_else((512u + groups - 1u) / groups);
    if (sl < 1u) sl = 1u;
    if (sl > ntiles) sl = ntiles;
    if (sl > 64u) sl = 64u;
    const uint32_t per = (ntiles + sl - 1u) / sl;
    *slices = (ntiles + per - 1u) / per;  // no empty slice
    *j_chunk = per * tile;
}
// the slices' rows, then the per-1024-record velocity sums of the whole set, then the step's flag word
size_t boids_split_rows_bytes(uint32_t count, uint32_t slices) { return (size_t)slices * count * 3u * sizeof(float4); }
size_t boids_split_bytes(uint32_t n_total, uint32_t count, uint32_t slices)
{
    return boids_split_rows_bytes(count, slices) + (size_t)((n_total + 1023u) / 1024u) * sizeof(float4) + 64u;
}
void boids_split_pointers(nbk::BoidsArgs *a, void *scratch, uint32_t slices, uint32_t chunk)
{
    a->partial = (float4 *)scratch;
    a->j_chunk = chunk;
    a->vtot = (float4 *)((char *)scratch + boids_split_rows_bytes(a->count, slices));
    a->gflags = (uint32_t *)(a->vtot + (a->n_total + 1023u) / 1024u);
}

// Launch-API calls run on the device that owns the caller's buffers, not on whatever device happens to be current in
// this thread (one process per GPU normally makes them equal; a host that forgot hipSetDevice would otherwise launch on
// device 0 against another device's memory).
int select_device_of(const void *dev_ptr, std::string *err)
{
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, dev_ptr);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return NB_OK;  // not a pointer the runtime knows (e.g. a test's fake address): leave the current device alone
    }
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur != attr.device) {
        e = hipSetDevice(attr.device);
        if (e != hipSuccess) {
            *err = std::string("nb: hipSetDevice failed: ") + hipGetErrorString(e);
            return NB_ERR_HIP;
        }
    }
    return NB_OK;
}

uint64_t splitmix64(uint64_t &s)
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

float uniform_f32(uint64_t &s, float lo, float hi)
{
    const uint32_t bits24 = (uint32_t)(splitmix64(s) >> 40);
    const float u = (float)bits24 * 0x1.0p-24f;
    const float scale = hi - lo;
    const float v = scale * u;
    return lo + v;
}

}  // namespace

struct nb_ctx {
    uint32_t n = 0;
    nb_params p{};
    Plan plan{};
    hipStream_t stream = nullptr;
    float4 *pos[2] = {nullptr, nullptr};
    float4 *vel = nullptr;
    float4 *vel_alt = nullptr;  // second velocity buffer: boids reads every old velocity, so velocities ping-pong too
    float *stage = nullptr;   // 3n floats: stride-3 staging for upload/download
    float4 *inst = nullptr;   // 4n float4, allocated on first use
    float4 *cams = nullptr;   // 4n float4, allocated on first use (nb_cameras)
    float *xfer = nullptr;    // 22n floats [matrices 16n | positions 3n | velocities 3n]: one-copy round trip of the drop-in calls
    float *hxfer = nullptr;   // its pinned host twin
    float *hxfer_dev = nullptr;  // the device's address of hxfer (mapped host memory: kernels of the small-set drop-in read and write it directly)
    uint32_t *done_counter = nullptr;  // export_kernel's workgroup counter (ExportDone)
    uint32_t done_seq = 0, polled_calls = 0;
    uint32_t poll_holdoff = 0, poll_backoff = 0;  // calls that take the stream wait after a poll that timed out (wait_export)
    void *scratch = nullptr;
    StatusWord status;        // sticky failure word of this context's block-chain launches
    int cur = 0;
    bool uploaded = false;
    uint64_t steps = 0;
    std::string err;
};

#define NB_HIP(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            (ctx)->err = std::string("nb: " #call " failed: ") + hipGetErrorString(e_);            \
            return (e_ == hipErrorOutOfMemory) ? NB_ERR_ALLOC : NB_ERR_HIP;                        \
        }                                                                                          \
    } while (0)

NB_EXPORT int nb_abi_version(void) { return NB_ABI_VERSION; }

NB_EXPORT void nb_default_params(nb_params *p)
{
    if (!p) return;
    p->dt = 0.1f;          // main.rs:411
    p->G = 0.001f;         // main.rs:412
    p->bias = 0.0000001f;  // main.rs:413
    p->tile = 0;
    p->mode = NB_MODE_STRICT;
}

NB_EXPORT int nb_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        g_tls_error = std::string("nb: hipGetDeviceCount failed: ") + hipGetErrorString(e);
        return NB_ERR_NO_DEVICE;
    }
    return n;
}

NB_EXPORT const char *nb_last_error(const nb_ctx *ctx) { return ctx ? ctx->err.c_str() : g_tls_error.c_str(); }

NB_EXPORT int nb_init_state(uint64_t seed, uint32_t n, float *pos_xyz, float *vel_xyz)
{
    if (!pos_xyz || !vel_xyz) {
        g_tls_error = "nb_init_state: null array";
        return NB_ERR_INVALID;
    }
    uint64_t s = seed;
    for (uint32_t i = 0; i < n; ++i) {  // main.rs:738-742
        vel_xyz[3 * (size_t)i + 0] = uniform_f32(s, -0.0f, 0.1f);
        vel_xyz[3 * (size_t)i + 1] = uniform_f32(s, -0.0f, 0.1f);
        vel_xyz[3 * (size_t)i + 2] = 0.0f;
    }
    for (uint32_t i = 0; i < n; ++i) {  // main.rs:743-747
        pos_xyz[3 * (size_t)i + 0] = uniform_f32(s, -100.0f, 100.0f);
        pos_xyz[3 * (size_t)i + 1] = uniform_f32(s, -100.0f, 100.0f);
        pos_xyz[3 * (size_t)i + 2] = 0.0f;
    }
    return NB_OK;
}

NB_EXPORT void nb_destroy(nb_ctx *ctx)
{
    if (!ctx) return;
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->pos[0]) (void)hipFree(ctx->pos[0]);
    if (ctx->pos[1]) (void)hipFree(ctx->pos[1]);
    if (ctx->vel) (void)hipFree(ctx->vel);
    if (ctx->vel_alt) (void)hipFree(ctx->vel_alt);
    if (ctx->stage) (void)hipFree(ctx->stage);
    if (ctx->inst) (void)hipFree(ctx->inst);
    if (ctx->cams) (void)hipFree(ctx->cams);
    if (ctx->xfer) (void)hipFree(ctx->xfer);
    if (ctx->hxfer) (void)hipHostFree(ctx->hxfer);
    if (ctx->done_counter) (void)hipFree(ctx->done_counter);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->status.w) (void)hipFree(ctx->status.w);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

// Sets up to this size cross the bus as ONE pinned copy each way (they are all latency: every separate copy and wait
// shows); larger sets are all bandwidth and copy straight from / to the caller's arrays.  Measured: tools/crossover.py.
constexpr uint32_t kRoundtripMax = 16384;
// Small sets: the pack / unpack kernels read and write the pinned host buffer through the bus themselves instead of a DMA copy
// each way -- 7-14 us less per call at every size up to kRoundtripMax (tools/dropin_small.py, profiles/r02/dropin_small.log:
// N = 100 32.6 -> 25.6 us, N = 2 048 66.7 -> 52.8, N = 16 384 240 -> 226).  NB_DROPIN_ZERO_COPY=0 stages through DMA copies.
constexpr uint32_t kZeroCopyMax = kRoundtripMax;
static bool small_set_zero_copy(uint32_t n) { return overrides().dropin_zero_copy.or_else(n <= kZeroCopyMax ? 1u : 0u) != 0u; }

// after a small-set export whose kernel copied the status word behind the 22n floats of hxfer (and the wait)
static int status_from_tail(nb_ctx *c, bool with_status)
{
    if (!with_status) return NB_OK;
    uint32_t v;
    std::memcpy(&v, c->hxfer + 22 * (size_t)c->n, sizeof(v));
    if (v == 0u) {
        c->status.dirty = false;
        return NB_OK;
    }
    return check_status(&c->status, &c->err);  // reads it again, clears it, reports NB_ERR_STATE
}

// The smallest sets are ALL latency, the wait included: hipStreamSynchronize returns ~4 us after a kernel's last store has reached
// the host (tools/ubench_sync.hip, profiles/r04/ubench_sync.log: 12.5 against 8.3 us for launch + wait of an empty kernel).  So
// where the export kernel writes mapped host memory anyway, its last workgroup also writes a sequence number behind the
// results and the host polls that word: per call 28.4 -> 24.4 us at N = 100, 52.8 -> 49.6 at 2 048 (profiles/r04/dropin_small.log).
// Up to kPollMax bodies -- the reference's own ceiling (main.rs:653); from 4 096 bodies on the system-scope fences of the many
// workgroups cost more than the wait saves (72.6 -> 77.3 us).  A word that does not arrive within kPollBudget falls back to the
// stream wait, which also reports what went wrong.  NB_DROPIN_POLL=0: always the stream wait; =1: polled up to kZeroCopyMax.
constexpr uint32_t kPollMax = 2048;
constexpr size_t kDoneWordAt = 8;  // floats behind the 22n results (the status word is at 0)
static bool small_set_poll(uint32_t n) { return small_set_zero_copy(n) && overrides().dropin_poll.or_else(n <= kPollMax ? 1u : 0u) != 0u; }

static int ensure_xfer(nb_ctx *c)
{
    const size_t bytes = ((size_t)c->n * 22 + 16) * sizeof(float);  // + a tail: the status word and the completion word ride along
    if (!c->xfer) NB_HIP(c, hipMalloc((void **)&c->xfer, bytes));
    if (!c->hxfer) {
        NB_HIP(c, hipHostMalloc((void **)&c->hxfer, bytes, hipHostMallocMapped));
        NB_HIP(c, hipHostGetDevicePointer((void **)&c->hxfer_dev, c->hxfer, 0));
        std::memset(c->hxfer + 22 * (size_t)c->n, 0, 16 * sizeof(float));
    }
    if (!c->done_counter) {
        NB_HIP(c, hipMalloc((void **)&c->done_counter, sizeof(uint32_t)));
        NB_HIP(c, hipMemset(c->done_counter, 0, sizeof(uint32_t)));
    }
    return NB_OK;
}

// what export_kernel is to signal for this call (nothing unless the set is polled)
static nbk::ExportDone export_done(nb_ctx *c, bool polled)
{
    nbk::ExportDone d;
    if (polled && c->poll_holdoff) {  // the word kept the host spinning not long ago (wait_export): the stream wait for a while
        --c->poll_holdoff;
        polled = false;
    }
    if (polled) {
        d.counter = c->done_counter;
        d.word = (uint32_t *)(c->hxfer_dev + 22 * (size_t)c->n + kDoneWordAt);
        d.seq = ++c->done_seq ? c->done_seq : ++c->done_seq;  // never 0 (the word's initial value)
    }
    return d;
}

static inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    __asm__ __volatile__("yield");
#else
    std::this_thread::yield();
#endif
}

// waits for the export launched last: the polled word, or the stream.  The poll is ADAPTIVE: a word that does not arrive within
// kPollBudget (the GPU is busy with something else -- the reference renders through the same device every frame, so the export
// kernel may routinely queue behind graphics work) costs that call the spin AND the stream wait, so the next 64 calls of the
// context go straight to the stream wait, twice as many after every further timeout (up to 4 096); a poll that succeeds
// resets the back-off.  The fallback also re-zeroes the workgroup counter: only the LAST workgroup of an export launch resets it,
// so a launch that died half way would leave every later polled call waiting for a count that never comes.
static int wait_export(nb_ctx *c, const nbk::ExportDone &d)
{
    if (d.word) {
        const volatile uint32_t *const w = (const volatile uint32_t *)(c->hxfer + 22 * (size_t)c->n + kDoneWordAt);
        const auto kPollBudget = std::chrono::microseconds(overrides().dropin_poll_budget_us.or_else(2000u));  // (the knob: tests)
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t spins = 1;; ++spins) {
            if (*w == d.seq) {
                std::atomic_thread_fence(std::memory_order_acquire);
                c->poll_backoff = 0;
                // (now and then the runtime gets to see the stream idle: it retires its own bookkeeping at a wait)
                if ((++c->polled_calls & 1023u) == 0u) NB_HIP(c, hipStreamSynchronize(c->stream));
                return NB_OK;
            }
            if ((spins & 255u) == 0u && std::chrono::steady_clock::now() - t0 > kPollBudget) break;
            cpu_relax();
        }
        c->poll_backoff = c->poll_backoff ? std::min(c->poll_backoff * 2u, 4096u) : 64u;
        c->poll_holdoff = c->poll_backoff;
        NB_HIP(c, hipStreamSynchronize(c->stream));
        if (*w != d.seq) NB_HIP(c, hipMemsetAsync(c->done_counter, 0, sizeof(uint32_t), c->stream));  // (stream-ordered in front of the next export)
        return NB_OK;
    }
    NB_HIP(c, hipStreamSynchronize(c->stream));
    return NB_OK;
}

static int create_impl(nb_ctx *c)
{
    const size_t rec = (size_t)c->n * sizeof(float4);
    NB_HIP(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    NB_HIP(c, hipMalloc((void **)&c->pos[0], rec));
    NB_HIP(c, hipMalloc((void **)&c->pos[1], rec));
    NB_HIP(c, hipMalloc((void **)&c->vel, rec));
    NB_HIP(c, hipMalloc((void **)&c->stage, (size_t)c->n * 3 * sizeof(float)));
    const size_t sb = plan_scratch_bytes(c->plan, c->n);
    if (sb) NB_HIP(c, hipMalloc(&c->scratch, sb));
    return NB_OK;
}

NB_EXPORT int nb_create(uint32_t n, uint32_t n_devices, const nb_params *params, nb_ctx **out)
{
    if (!out) {
        g_tls_error = "nb_create: out is null";
        return NB_ERR_INVALID;
    }
    *out = nullptr;
    if (n_devices != 1) {
        g_tls_error =
            "nb_create: n_devices must be 1 (one process drives one GPU; shard with nb_launch_step and an all-gather)";
        return NB_ERR_UNSUPPORTED;
    }
    nb_params p;
    if (params)
        p = *params;
    else
        nb_default_params(&p);
    Plan plan;
    int rc = make_plan(p, n, n, &plan, &g_tls_error);
    if (rc != NB_OK) return rc;
    rc = check_device(&g_tls_error);
    if (rc != NB_OK) return rc;
    nb_ctx *c = new (std::nothrow) nb_ctx();
    if (!c) {
        g_tls_error = "nb_create: out of host memory";
        return NB_ERR_ALLOC;
    }
    c->n = n;
    c->p = p;
    c->plan = plan;
    rc = create_impl(c);
    if (rc != NB_OK) {
        g_tls_error = c->err;
        nb_destroy(c);
        return rc;
    }
    *out = c;
    return NB_OK;
}

NB_EXPORT int nb_upload(nb_ctx *ctx, const float *pos_xyz, const float *vel_xyz)
{
    if (!ctx) {
        g_tls_error = "nb_upload: ctx is null";
        return NB_ERR_INVALID;
    }
    if (!pos_xyz || !vel_xyz) {
        ctx->err = "nb_upload: null array";
        return NB_ERR_INVALID;
    }
    const size_t bytes = (size_t)ctx->n * 3 * sizeof(float);
    ctx->cur = 0;
    if (ctx->n <= kRoundtripMax) {  // one pinned copy, no wait: the stream orders it before the first step
        int rc = ensure_xfer(ctx);
        if (rc != NB_OK) return rc;
        NB_HIP(ctx, hipStreamSynchronize(ctx->stream));  // a download still reading hxfer must be done (it always is: downloads wait)
        std::memcpy(ctx->hxfer, pos_xyz, bytes);
        std::memcpy(ctx->hxfer + 3 * (size_t)ctx->n, vel_xyz, bytes);
        float *src = ctx->hxfer_dev;  // the kernel reads the pinned buffer through the bus ...
        if (!small_set_zero_copy(ctx->n)) {  // ... or a DMA copy stages it in device memory first
            NB_HIP(ctx, hipMemcpyAsync(ctx->xfer, ctx->hxfer, 2 * bytes, hipMemcpyHostToDevice, ctx->stream));
            src = ctx->xfer;
        }
        NB_HIP(ctx, nbk::launch_import(ctx->n, src, src + 3 * (size_t)ctx->n, ctx->pos[0], ctx->vel, ctx->stream));
        NB_HIP(ctx, hipStreamSynchronize(ctx->stream));  // hxfer / xfer are reused by the next call
        ctx->uploaded = true;
        ctx->steps = 0;
        return NB_OK;
    }
    NB_HIP(ctx, hipMemcpyAsync(ctx->stage, pos_xyz, bytes, hipMemcpyHostToDevice, ctx->stream));
    NB_HIP(ctx, nbk::launch_pack(ctx->n, ctx->stage, ctx->pos[0], ctx->stream));
    NB_HIP(ctx, hipStreamSynchronize(ctx->stream));  // stage is reused; the host array is not retained
    NB_HIP(ctx, hipMemcpyAsync(ctx->stage, vel_xyz, bytes, hipMemcpyHostToDevice, ctx->stream));
    NB_HIP(ctx, nbk::launch_pack(ctx->n, ctx->stage, ctx->vel, ctx->stream));
    NB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->uploaded = true;
    ctx->steps = 0;
    return NB_OK;
}

NB_EXPORT int nb_step(nb_ctx *ctx, uint32_t k)
{
    if (!ctx) {
        g_tls_error = "nb_step: ctx is null";
        return NB_ERR_INVALID;
    }
    if (!ctx->uploaded) {
        ctx->err = "nb_step: no state uploaded (call nb_upload first)";
        return NB_ERR_STATE;
    }
    RoctxRange range("nb_step");
    for (uint32_t s = 0; s < k; ++s) {
        int rc = launch_step_planned(ctx->p, ctx->plan, ctx->n, 0, ctx->n, ctx->pos[ctx->cur], ctx->pos[ctx->cur ^ 1],
                                     ctx->vel, ctx->scratch, ctx->stream, &ctx->err, &ctx->status);
        if (rc != NB_OK) return rc;
        ctx->cur ^= 1;
        ctx->steps++;
    }
    return NB_OK;
}

NB_EXPORT void nb_boids_default_params(nb_boids_params *p)
{
    if (!p) return;
    p->dt = 0.04f;                // main.rs:450
    p->rule_1_distance = 1000.0f; // main.rs:451
    p->rule_2_distance = 5.0f;    // main.rs:452
    p->rule_3_distance = 500.0f;  // main.rs:453
    p->rule_1_scale = 0.02f;      // main.rs:454
    p->rule_2_scale = 0.05f;      // main.rs:455
    p->rule_3_scale = 0.5f;       // main.rs:456
    p->tile = 0;
}

NB_EXPORT int nb_step_boids(nb_ctx *ctx, uint32_t k, const nb_boids_params *params)
{
    if (!ctx) {
        g_tls_error = "nb_step_boids: ctx is null";
        return NB_ERR_INVALID;
    }
    if (!ctx->uploaded) {
        ctx->err = "nb_step_boids: no state uploaded (call nb_upload first)";
        return NB_ERR_STATE;
    }
    nb_boids_params p;
    if (params)
        p = *params;
    else
        nb_boids_default_params(&p);
    nbk::BoidsArgs a;
    uint32_t tile = 0;
    int rc = make_boids_args(p, ctx->n, 0, ctx->n, &a, &tile, &ctx->err);
    if (rc != NB_OK) return rc;
    if (!ctx->vel_alt) NB_HIP(ctx, hipMalloc((void **)&ctx->vel_alt, (size_t)ctx->n * sizeof(float4)));
    RoctxRange range("nb_step_boids");
    for (uint32_t s = 0; s < k; ++s) {
        a.pos_in = ctx->pos[ctx->cur];
        a.pos_out = ctx->pos[ctx->cur ^ 1];
        a.vel_in = ctx->vel;
        a.vel_out = ctx->vel_alt;
        NB_HIP(ctx, nbk::launch_boids(a, tile, boids_form(a.count), ctx->stream));
        ctx->cur ^= 1;
        std::swap(ctx->vel, ctx->vel_alt);
        ctx->steps++;
    }
    return NB_OK;
}

NB_EXPORT int nb_step_random(nb_ctx *ctx, uint32_t k, uint64_t seed)
{
    if (!ctx) {
        g_tls_error = "nb_step_random: ctx is null";
        return NB_ERR_INVALID;
    }
    if (!ctx->uploaded) {
        ctx->err = "nb_step_random: no state uploaded (call nb_upload first)";
        return NB_ERR_STATE;
    }
    for (uint32_t s = 0; s < k; ++s) {
        NB_HIP(ctx, nbk::launch_random(0, ctx->n, ctx->pos[ctx->cur], ctx->vel, seed, ctx->steps, ctx->stream));
        ctx->steps++;
    }
    return NB_OK;
}

NB_EXPORT int nb_device_state(nb_ctx *ctx, const void **pos_rec, const void **vel_rec, const void **inst_16n)
{
    if (!ctx) {
        g_tls_error = "nb_device_state: ctx is null";
        return NB_ERR_INVALID;
    }
    if (!ctx->uploaded) {
        ctx->err = "nb_device_state: no state uploaded";
        return NB_ERR_STATE;
    }
    if (inst_16n) {
        if (!ctx->inst) NB_HIP(ctx, hipMalloc((void **)&ctx->inst, (size_t)ctx->n * 16 * sizeof(float)));
        NB_HIP(ctx, nbk::launch_instances(ctx->n, ctx->pos[ctx->cur], ctx->vel, ctx->inst, ctx->stream, overrides().inst_device_libm.on() ? 1u : 0u));
        *inst_16n = ctx->inst;
    }
    if (pos_rec) *pos_rec = ctx->pos[ctx->cur];
    if (vel_rec) *vel_rec = ctx->vel;
    return NB_OK;
}

// The constant of a camera array: OPENGL_TO_WGPU_MATRIX * perspective(vertical_fov, aspect, near, far) (src/gfx.rs:12-17, 365,
// 367).  Host arithmetic only (one tanf per ARRAY, not per entity): cgmath 0.17's Deg -> Rad (deg * PI/180, the constant
// formed in binary64), PerspectiveFov -> Matrix4 (f = 1 / tan(fovy / 2); f / aspect, f, (far + near) / (near - far), -1,
// (2 far near) / (near - far)) and Matrix4 * Matrix4 (((a r0 + b r1) + c r2) + d r3 per entry), its assertions as an error.
NB_EXPORT int nb_camera_constant(float vertical_fov_deg, float aspect_ratio, float near_plane, float far_plane, float *cp16)
{
    if (!cp16) {
        g_tls_error = "nb_camera_constant: cp16 is null";
        return NB_ERR_INVALID;
    }
    const float pi = (float)3.14159265358979323846;
    const float fovy = vertical_fov_deg * (float)(3.14159265358979323846 / 180.0);
    if (!(fovy > 0.0f) || !(fovy < pi) || !(aspect_ratio != 0.0f) || !(near_plane > 0.0f) || !(far_plane > 0.0f) || !(far_plane != near_plane)) {
        g_tls_error = "nb_camera_constant: cgmath::perspective asserts 0 < fovy < 180 degrees, aspect != 0, near > 0, far > 0, far != near";
        return NB_ERR_INVALID;
    }
    const float f = 1.0f / std::tan(fovy / 2.0f);
    float proj[16] = {0.0f};  // column-major
    proj[0] = f / aspect_ratio;
    proj[5] = f;
    proj[10] = (far_plane + near_plane) / (near_plane - far_plane);
    proj[11] = -1.0f;
    proj[14] = ((2.0f * far_plane) * near_plane) / (near_plane - far_plane);
    const float corr[16] = {1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.5f, 0.0f, 0.0f, 0.0f, 0.5f, 1.0f};  // src/gfx.rs:12-17
    for (int col = 0; col < 4; ++col)
        for (int row = 0; row < 4; ++row) {
            const float t0 = corr[row] * proj[4 * col], t1 = corr[4 + row] * proj[4 * col + 1];
            const float t2 = corr[8 + row] * proj[4 * col + 2], t3 = corr[12 + row] * proj[4 * col + 3];
            cp16[4 * col + row] = ((t0 + t1) + t2) + t3;
        }
    return NB_OK;
}

NB_EXPORT int nb_cameras(nb_ctx *ctx, const float *up_xyz, const float *cp16, float *out_16n)
{
    if (!ctx) {
        g_tls_error = "nb_cameras: ctx is null";
        return NB_ERR_INVALID;
    }
    if (!up_xyz || !cp16 || !out_16n) {
        ctx->err = "nb_cameras: null argument";
        return NB_ERR_INVALID;
    }
    if (!ctx->uploaded) {
        ctx->err = "nb_cameras: no state uploaded";
        return NB_ERR_STATE;
    }
    if (!ctx->cams) NB_HIP(ctx, hipMalloc((void **)&ctx->cams, (size_t)ctx->n * 16 * sizeof(float)));
    NB_HIP(ctx, nbk::launch_cameras(ctx->n, ctx->pos[ctx->cur], ctx->vel, up_xyz, cp16, ctx->cams, ctx->stream));
    NB_HIP(ctx, hipMemcpyAsync(out_16n, ctx->cams, (size_t)ctx->n * 16 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    NB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NB_OK;
}

NB_EXPORT int nb_sync(nb_ctx *ctx)
{
    if (!ctx) {
        g_tls_error = "nb_sync: ctx is null";
        return NB_ERR_INVALID;
    }
    NB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return check_status(&ctx->status, &ctx->err);
}

NB_EXPORT uint64_t nb_steps_done(const nb_ctx *ctx) { return ctx ? ctx->steps : 0; }

NB_EXPORT int nb_download(nb_ctx *ctx, float *pos_xyz, float *vel_xyz, float *inst_16n)
{
    if (!ctx) {
        g_tls_error = "nb_download: ctx is null";
        return NB_ERR_INVALID;
    }
    if (!ctx->uploaded) {
        ctx->err = "nb_download: no state uploaded";
        return NB_ERR_STATE;
    }
    const size_t bytes = (size_t)ctx->n * 3 * sizeof(float);
    if (ctx->n <= kRoundtripMax && (pos_xyz || vel_xyz || inst_16n)) {  // [matrices 16n | positions 3n | velocities 3n], one copy, one wait
        int rc = ensure_xfer(ctx);
        if (rc != NB_OK) return rc;
        const size_t n = ctx->n;
        const bool zero_copy = small_set_zero_copy(ctx->n);
        float *const dst = zero_copy ? ctx->hxfer_dev : ctx->xfer;
        // the sticky status word of this context's block-chain launches comes home in the same buffer: no second wait
        const bool with_status = ctx->status.dirty && ctx->status.w;
        const nbk::ExportDone done = export_done(ctx, small_set_poll(ctx->n));
        NB_HIP(ctx, nbk::launch_export(ctx->n, ctx->pos[ctx->cur], ctx->vel, inst_16n ? (float4 *)dst : nullptr,
                                       pos_xyz ? dst + 16 * n : nullptr, vel_xyz ? dst + 19 * n : nullptr,
                                       with_status ? ctx->status.w : nullptr, with_status ? (uint32_t *)(dst + 22 * n) : nullptr, ctx->stream,
                                       overrides().inst_device_libm.on() ? 1u : 0u, done));
        if (!zero_copy) {
            const size_t lo = inst_16n ? 0 : (pos_xyz ? 16 * n : 19 * n), hi = with_status ? 22 * n + 1 : vel_xyz ? 22 * n : (pos_xyz ? 19 * n : 16 * n);
            NB_HIP(ctx, hipMemcpyAsync(ctx->hxfer + lo, ctx->xfer + lo, (hi - lo) * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        }
        rc = wait_export(ctx, done);
        if (rc != NB_OK) return rc;
        if (inst_16n) std::memcpy(inst_16n, ctx->hxfer, n * 16 * sizeof(float));
        if (pos_xyz) std::memcpy(pos_xyz, ctx->hxfer + 16 * n, bytes);
        if (vel_xyz) std::memcpy(vel_xyz, ctx->hxfer + 19 * n, bytes);
        return status_from_tail(ctx, with_status);
    }
    if (pos_xyz) {
        NB_HIP(ctx, nbk::launch_unpack(ctx->n, ctx->pos[ctx->cur], ctx->stage, ctx->stream));
        NB_HIP(ctx, hipMemcpyAsync(pos_xyz, ctx->stage, bytes, hipMemcpyDeviceToHost, ctx->stream));
        NB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (vel_xyz) {
        NB_HIP(ctx, nbk::launch_unpack(ctx->n, ctx->vel, ctx->stage, ctx->stream));
        NB_HIP(ctx, hipMemcpyAsync(vel_xyz, ctx->stage, bytes, hipMemcpyDeviceToHost, ctx->stream));
        NB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (inst_16n) {
        if (!ctx->inst) NB_HIP(ctx, hipMalloc((void **)&ctx->inst, (size_t)ctx->n * 16 * sizeof(float)));
        NB_HIP(ctx, nbk::launch_instances(ctx->n, ctx->pos[ctx->cur], ctx->vel, ctx->inst, ctx->stream, overrides().inst_device_libm.on() ? 1u : 0u));
        NB_HIP(ctx, hipMemcpyAsync(inst_16n, ctx->inst, (size_t)ctx->n * 16 * sizeof(float), hipMemcpyDeviceToHost,
                                   ctx->stream));
        NB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (!pos_xyz && !vel_xyz && !inst_16n) NB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return check_status(&ctx->status, &ctx->err);
}

#include "nb_dropin.inc"

#include "nb_selftest.inc"

// ---- launch API -----------------------------------------------------------------------------------------

NB_EXPORT size_t nb_scratch_bytes(const nb_params *params, uint32_t n_total, uint32_t count)
{
    nb_params p;
    if (params)
        p = *params;
    else
        nb_default_params(&p);
    Plan pl;
    std::string err;
    if (make_plan(p, n_total, count, &pl, &err) != NB_OK) return 0;
    return plan_scratch_bytes(pl, count);
}

NB_EXPORT int nb_launch_step(const nb_params *params, uint32_t n_total, uint32_t first, uint32_t count, const void *pos_in,
                             void *pos_out, void *vel, void *scratch, size_t scratch_bytes, void *stream)
{
    nb_params p;
    if (params)
        p = *params;
    else
        nb_default_params(&p);
    if (!pos_in || !pos_out || !vel || pos_in == pos_out) {
        g_tls_error = "nb_launch_step: pos_in, pos_out, vel must be non-null and pos_out must not alias pos_in";
        return NB_ERR_INVALID;
    }
    if ((uint64_t)first + (uint64_t)count > (uint64_t)n_total) {
        g_tls_error = "nb_launch_step: [first, first+count) exceeds n_total";
        return NB_ERR_INVALID;
    }
    const Plan *pl = nullptr;
    int rc = cached_plan(p, n_total, count, &pl, &g_tls_error);
    if (rc != NB_OK) return rc;
    const size_t need = plan_scratch_bytes(*pl, count);
    if (need && (!scratch || scratch_bytes < need)) {
        g_tls_error = "nb_launch_step: scratch smaller than nb_scratch_bytes()";
        return NB_ERR_INVALID;
    }
    rc = check_device(&g_tls_error);
    if (rc != NB_OK) return rc;
    if (!stream) {  // a caller that brings a stream has made that stream's device current; the NULL stream says nothing
        rc = select_device_of(pos_in, &g_tls_error);
        if (rc != NB_OK) return rc;
    }
    return launch_step_planned(p, *pl, n_total, first, count, pos_in, pos_out, vel, scratch, (hipStream_t)stream, &g_tls_error);
}

NB_EXPORT int nb_launch_status(void *stream)
{
    int rc = check_device(&g_tls_error);
    if (rc != NB_OK) return rc;
    // the status word is per device: ask about the device the stream belongs to, not whichever is current -- and leave the
    // caller's current device as it was (a host that drives several GPUs from one thread goes on allocating where it was)
    int cur = -1, restore = -1;
    if (stream) {
        hipDevice_t sdev = -1;
        if (hipStreamGetDevice((hipStream_t)stream, &sdev) == hipSuccess && hipGetDevice(&cur) == hipSuccess && sdev >= 0 && sdev != cur) {
            if (hipSetDevice(sdev) == hipSuccess) restore = cur;
        } else {
            (void)hipGetLastError();
        }
    }
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) {
        g_tls_error = std::string("nb_launch_status: hipStreamSynchronize failed: ") + hipGetErrorString(e);
        rc = NB_ERR_HIP;
    } else {
        rc = check_device_status(&g_tls_error);
    }
    if (restore >= 0) (void)hipSetDevice(restore);
    return rc;
}

// ---- a FAST step in two phases ("Overlap") -------------------------------------------------
namespace {
struct PhasePlan {
    Plan base;                 // tile, ib, the sharing verdict, force_3d ...
    uint32_t len[2];           // records folded by phase 0 (the range) and phase 1 (the rest)
    uint32_t groups[2], slices[2], chunk[2];
    uint32_t fsl;              // 1 = the phases fold through scalar loads (step_fast_sl_kernel; the planes area in front of the rows), 0 = LDS forms
};

int make_phase_plan(const nb_params &p, uint32_t n_total, uint32_t count, uint32_t j_lo, uint32_t j_hi, PhasePlan *out, std::string *err)
{
    if (p.mode != NB_MODE_FAST) {
        *err = "nb: a step in phases is FAST only: the reference's sum runs j = 0..N-1 in order, and STRICT keeps that order";
        return NB_ERR_UNSUPPORTED;
    }
    if (j_lo > j_hi || j_hi > n_total) {
        *err = "nb: need j_lo <= j_hi <= n_total";
        return NB_ERR_INVALID;
    }
    int rc = make_plan(p, n_total, count, &out->base, err);
    if (rc != NB_OK) return rc;
    out->len[0] = j_hi - j_lo;
    out->len[1] = n_total - out->len[0];
    // Through scalar loads where a shard's one-call step would go that way (the plan's own choice, never the pairs form: a phase
    // folds a range of j's, not pairs): eight waves per workgroup, chunks of whole 256-record tiles; else the LDS forms.
    out->fsl = (out->base.fsl || out->base.pairs) ? 1u : 0u;
    for (int ph = 0; ph < 2; ++ph) {
        uint32_t tile = out->fsl ? 256u : out->base.tile;
        fast_split(out->len[ph], count, out->base.ib, out->fsl ? 8u : out->base.waves, &tile, &out->groups[ph], &out->slices[ph], &out->chunk[ph]);
    }
    return NB_OK;
}
size_t phase_scratch_bytes(const PhasePlan &pp, uint32_t count)
{
    return (pp.fsl ? nbk::strict_bc_scratch_bytes(pp.base.n_total) : 0) + (size_t)(pp.slices[0] + pp.slices[1]) * count * sizeof(float4);
}
}  // namespace

namespace {
// as cached_plan: a rank asks for the same two phases every step
int cached_phase_plan(const nb_params &p, uint32_t n_total, uint32_t count, uint32_t j_lo, uint32_t j_hi, const PhasePlan **out,
                      std::string *err)
{
    struct Entry {
        bool valid = false;
        nb_params p{};
        uint32_t n_total = 0, count = 0, j_lo = 0, j_hi = 0, generation = 0;
        PhasePlan pp{};
    };
    constexpr int kEntries = 2;
    thread_local Entry cache[kEntries];
    thread_local int next = 0;
    const uint32_t gen = overrides().generation;
    for (int i = 0; i < kEntries; ++i) {
        const Entry &e = cache[i];
        if (e.valid && e.n_total == n_total && e.count == count && e.j_lo == j_lo && e.j_hi == j_hi && e.generation == gen &&
            std::memcmp(&e.p, &p, sizeof(p)) == 0) {
            *out = &e.pp;
            return NB_OK;
        }
    }
    PhasePlan pp;
    int rc = make_phase_plan(p, n_total, count, j_lo, j_hi, &pp, err);
    if (rc != NB_OK) return rc;
    Entry &e = cache[next];
    next = (next + 1) % kEntries;
    e.valid = true;
    e.p = p;
    e.n_total = n_total;
    e.count = count;
    e.j_lo = j_lo;
    e.j_hi = j_hi;
    e.generation = gen;
    e.pp = pp;
    *out = &e.pp;
    return NB_OK;
}
}  // namespace

NB_EXPORT size_t nb_scratch_bytes_phased(const nb_params *params, uint32_t n_total, uint32_t count, uint32_t j_lo, uint32_t j_hi)
{
    nb_params p;
    if (params)
        p = *params;
    else
        nb_default_params(&p);
    PhasePlan pp;
    std::string err;
    if (make_phase_plan(p, n_total, count, j_lo, j_hi, &pp, &err) != NB_OK) return 0;
    return phase_scratch_bytes(pp, count);
}

namespace {
// one phase of a FAST step on planned shapes (nb_launch_step_phase, nb_shard_step with the overlapped exchange)
int launch_phase_planned(const nb_params &p, const PhasePlan &pp, uint32_t n_total, uint32_t first, uint32_t count, uint32_t j_lo,
                         int phase, const void *pos_in, void *pos_out, void *vel, void *scratch, hipStream_t stream, std::string *err)
{
    const Plan &pl = pp.base;
    nbk::StepArgs a{};
    a.pos_in = (const float4 *)pos_in;
    a.pos_out = (float4 *)pos_out;
    a.vel = (float4 *)vel;
    char *const rows = (char *)scratch + (pp.fsl ? nbk::strict_bc_scratch_bytes(n_total) : 0);  // the partial rows: behind the planes area
    a.partial = (float4 *)rows;
    a.n_total = n_total;
    a.first = first;
    a.count = count;
    a.dt = p.dt;
    a.G = p.G;
    a.bias = p.bias;
    a.force_ieee = pl.force_ieee;
    a.force_3d = pl.force_3d;
    a.always_partial = 1;
    hipError_t e = hipSuccess;
    if (phase == NB_PHASE_RANGE) {  // records [j_lo, j_lo + len[0]) -> partial rows [0, slices[0])
        a.j_count = pp.len[0];
        a.j_base = j_lo;
        a.hole_lo = 0xffffffffu;
        a.hole_len = 0;
        a.j_chunk = pp.chunk[0];
        a.partial_row0 = 0;
        if (pp.len[0])
            e = pp.fsl   ? nbk::launch_fast_sl_phase(a, pl.ib, pp.slices[0], true, j_lo, j_lo + pp.len[0], scratch, stream)
                : pl.waves ? nbk::launch_fast_wave(a, pl.tile, pl.ib, pl.waves, pp.slices[0], stream)
                           : nbk::launch_fast(a, pl.tile, pl.ib, pp.groups[0], pp.slices[0], stream);
        else
            e = hipMemsetAsync(rows, 0, (size_t)pp.slices[0] * count * sizeof(float4), stream);
    } else {  // the set without that range -> rows [slices[0], slices[0] + slices[1]), then every row in order + integrate
        a.j_count = pp.len[1];
        a.j_base = 0;
        a.hole_lo = j_lo;
        a.hole_len = pp.len[0];
        a.j_chunk = pp.chunk[1];
        a.partial_row0 = pp.slices[0];
        if (pp.len[1])
            e = pp.fsl   ? nbk::launch_fast_sl_phase(a, pl.ib, pp.slices[1], false, 0, 0, scratch, stream)
                : pl.waves ? nbk::launch_fast_wave(a, pl.tile, pl.ib, pl.waves, pp.slices[1], stream)
                           : nbk::launch_fast(a, pl.tile, pl.ib, pp.groups[1], pp.slices[1], stream);
        else
            e = hipMemsetAsync(rows + (size_t)pp.slices[0] * count * sizeof(float4), 0, (size_t)pp.slices[1] * count * sizeof(float4), stream);
        if (e == hipSuccess) e = nbk::launch_integrate_partials(a, pp.slices[0] + pp.slices[1], stream);
    }
    if (e != hipSuccess) {
        *err = std::string("nb: kernel launch failed (step phase): ") + hipGetErrorString(e);
        return NB_ERR_HIP;
    }
    return NB_OK;
}
}  // namespace

NB_EXPORT int nb_launch_step_phase(const nb_params *params, uint32_t n_total, uint32_t first, uint32_t count, uint32_t j_lo,
                                   uint32_t j_hi, int phase, const void *pos_in, void *pos_out, void *vel, void *scratch,
                                   size_t scratch_bytes, void *stream)
{
    nb_params p;
    if (params)
        p = *params;
    else
        nb_default_params(&p);
    if (!pos_in || !pos_out || !vel || pos_in == pos_out || !scratch || (phase != NB_PHASE_RANGE && phase != NB_PHASE_REST)) {
        g_tls_error = "nb_launch_step_phase: pos_in, pos_out, vel, scratch must be non-null, pos_out must not alias pos_in, phase is "
                      "NB_PHASE_RANGE or NB_PHASE_REST";
        return NB_ERR_INVALID;
    }
    if ((uint64_t)first + (uint64_t)count > (uint64_t)n_total) {
        g_tls_error = "nb_launch_step_phase: [first, first+count) exceeds n_total";
        return NB_ERR_INVALID;
    }
    const PhasePlan *ppp = nullptr;
    int rc = cached_phase_plan(p, n_total, count, j_lo, j_hi, &ppp, &g_tls_error);
    if (rc != NB_OK) return rc;
    const PhasePlan &pp = *ppp;
    if (scratch_bytes < phase_scratch_bytes(pp, count)) {
        g_tls_error = "nb_launch_step_phase: scratch smaller than nb_scratch_bytes_phased()";
        return NB_ERR_INVALID;
    }
    rc = check_device(&g_tls_error);
    if (rc != NB_OK) return rc;
    if (!stream) {  // as nb_launch_step: the NULL stream says nothing about the device, the buffers do
        rc = select_device_of(pos_in, &g_tls_error);
        if (rc != NB_OK) return rc;
    }
    return launch_phase_planned(p, pp, n_total, first, count, j_lo, phase, pos_in, pos_out, vel, scratch, (hipStream_t)stream, &g_tls_error);
}

#include "nb_ring_api.inc"


// 1: this library holds every launch shape the diagnostic knobs can name (built with -DNB_LEGACY_FORMS: make legacy); 0: the product
// build, which holds the shapes make_plan reaches by itself
NB_EXPORT int nb_diag_legacy_forms(void)
{
#ifdef NB_LEGACY_FORMS
    return 1;
#else
    return 0;
#endif
}

NB_EXPORT int nb_diag_rccl_solo(int on)
{
    g_rccl_solo.store(on != 0, std::memory_order_relaxed);
    return NB_OK;
}

// The kernels one step of this shape launches, dominant one first, as the library itself plans it (bench.py labels its
// roofline with this instead of restating make_plan).
NB_EXPORT int nb_diag_plan(const nb_params *params, uint32_t n_total, uint32_t count, char *out, size_t out_bytes)
{
    if (!out || out_bytes == 0) {
        g_tls_error = "nb_diag_plan: out is null";
        return NB_ERR_INVALID;
    }
    nb_params p;
    if (params)
        p = *params;
    else
        nb_default_params(&p);
    Plan pl;
    int rc = make_plan(p, n_total, count, &pl, &g_tls_error);
    if (rc != NB_OK) return rc;
    std::string k;
    if (p.mode == NB_MODE_STRICT)
        k = pl.bc ? "step_strict_bc_kernel,planes_kernel" : pl.sl ? "step_strict_sl_kernel,planes_kernel" : pl.pc ? "step_strict_pc_kernel" : "step_strict_kernel";
    else if (pl.pairs)
        k = nbk::fast_pairs_chunk(pl.n_total, pl.pairs_chunk) >= pl.n_total ? "step_fast_pairs_kernel,planes_kernel,pairs_diag_kernel,pairs_integrate_kernel"
                                                                             : "step_fast_pairs_kernel,planes_kernel,pairs_diag_kernel,pairs_accumulate_kernel,pairs_finish_kernel";
    else if (pl.fsl)
        k = std::string("step_fast_sl_kernel,planes_kernel") + (pl.slices > 1 ? ",integrate_partials_kernel" : "");
    else
        k = std::string(pl.waves ? "step_fast_wave_kernel" : "step_fast_kernel") + (pl.slices > 1 ? ",integrate_partials_kernel" : "");
    if (k.size() + 1 > out_bytes) {
        g_tls_error = "nb_diag_plan: out too small";
        return NB_ERR_INVALID;
    }
    std::memcpy(out, k.c_str(), k.size() + 1);
    return NB_OK;
}

NB_EXPORT int nb_diag_enable_env(int on)
{
    std::lock_guard<std::mutex> lock(g_overrides_mu);
    g_env_enabled.store(on != 0, std::memory_order_relaxed);
    const DebugOverrides *old = g_overrides.load(std::memory_order_acquire);
    g_overrides.store(parse_overrides(old ? old->generation + 1u : 1u), std::memory_order_release);
    return NB_OK;
}

NB_EXPORT int nb_debug_reload_env(void)
{
    std::lock_guard<std::mutex> lock(g_overrides_mu);
    g_env_enabled.store(true, std::memory_order_relaxed);  // asking for a re-read is asking for the knobs
    const DebugOverrides *old = g_overrides.load(std::memory_order_acquire);
    g_overrides.store(parse_overrides(old ? old->generation + 1u : 1u), std::memory_order_release);
    return NB_OK;
}

NB_EXPORT int nb_launch_boids_step(const nb_boids_params *params, uint32_t n_total, uint32_t first, uint32_t count,
                                   const void *pos_in, const void *vel_in, void *pos_out, void *vel_out, void *stream)
{
    nb_boids_params p;
    if (params)
        p = *params;
    else
        nb_boids_default_params(&p);
    if (!pos_in || !vel_in || !pos_out || !vel_out || pos_in == pos_out || vel_in == vel_out) {
        g_tls_error = "nb_launch_boids_step: buffers must be non-null and the outputs must not alias the inputs";
        return NB_ERR_INVALID;
    }
    nbk::BoidsArgs a;
    uint32_t tile = 0;
    int rc = make_boids_args(p, n_total, first, count, &a, &tile, &g_tls_error);
    if (rc != NB_OK) return rc;
    rc = check_device(&g_tls_error);
    if (rc != NB_OK) return rc;
    rc = select_device_of(pos_in, &g_tls_error);
    if (rc != NB_OK) return rc;
    a.pos_in = (const float4 *)pos_in;
    a.vel_in = (const float4 *)vel_in;
    a.pos_out = (float4 *)pos_out;
    a.vel_out = (float4 *)vel_out;
    hipError_t e = nbk::launch_boids(a, tile, boids_form(a.count), (hipStream_t)stream);
    if (e != hipSuccess) {
        g_tls_error = std::string("nb: boids kernel launch failed: ") + hipGetErrorString(e);
        return NB_ERR_HIP;
    }
    return NB_OK;
}

NB_EXPORT size_t nb_boids_split_scratch_bytes(const nb_boids_params *params, uint32_t n_total, uint32_t count)
{
    nb_boids_params p;
    if (params)
        p = *params;
    else
        nb_boids_default_params(&p);
    nbk::BoidsArgs a;
    uint32_t tile = 0, slices = 0, chunk = 0;
    std::string err;
    if (count == 0u) return 0;  // a rank without bodies launches nothing
    if (make_boids_args(p, n_total, 0, count, &a, &tile, &err) != NB_OK) return 0;
    boids_split_shape(n_total, count, tile, &slices, &chunk);
    return boids_split_bytes(n_total, count, slices);
}

NB_EXPORT int nb_launch_boids_step_split(const nb_boids_params *params, uint32_t n_total, uint32_t first, uint32_t count,
                                         const void *pos_in, const void *vel_in, void *pos_out, void *vel_out, void *scratch,
                                         size_t scratch_bytes, void *stream)
{
    nb_boids_params p;
    if (params)
        p = *params;
    else
        nb_boids_default_params(&p);
    if (!pos_in || !vel_in || !pos_out || !vel_out || !scratch || pos_in == pos_out || vel_in == vel_out) {
        g_tls_error = "nb_launch_boids_step_split: buffers must be non-null and the outputs must not alias the inputs";
        return NB_ERR_INVALID;
    }
    nbk::BoidsArgs a;
    uint32_t tile = 0, slices = 0, chunk = 0;
    int rc = make_boids_args(p, n_total, first, count, &a, &tile, &g_tls_error);
    if (rc != NB_OK) return rc;
    boids_split_shape(n_total, count, tile, &slices, &chunk);
    if (scratch_bytes < boids_split_bytes(n_total, count, slices)) {
        g_tls_error = "nb_launch_boids_step_split: scratch smaller than nb_boids_split_scratch_bytes()";
        return NB_ERR_INVALID;
    }
    rc = check_device(&g_tls_error);
    if (rc != NB_OK) return rc;
    rc = select_device_of(pos_in, &g_tls_error);
    if (rc != NB_OK) return rc;
    a.pos_in = (const float4 *)pos_in;
    a.vel_in = (const float4 *)vel_in;
    a.pos_out = (float4 *)pos_out;
    a.vel_out = (float4 *)vel_out;
    boids_split_pointers(&a, scratch, slices, chunk);
    hipError_t e = nbk::launch_boids_split(a, tile, slices, (hipStream_t)stream);
    if (e != hipSuccess) {
        g_tls_error = std::string("nb: boids kernel launch failed (split form): ") + hipGetErrorString(e);
        return NB_ERR_HIP;
    }
    return NB_OK;
}

#define NB_LAUNCH_TLS(call)                                                                  \
    do {                                                                                     \
        int rc_ = check_device(&g_tls_error);                                                \
        if (rc_ != NB_OK) return rc_;                                                        \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            g_tls_error = std::string("nb: " #call " failed: ") + hipGetErrorString(e_);     \
            return NB_ERR_HIP;                                                               \
        }                                                                                    \
        return NB_OK;                                                                        \
    } while (0)

NB_EXPORT int nb_launch_instances(uint32_t count, const void *pos, const void *vel, void *inst_16n, void *stream)
{
    if (!count || !pos || !vel || !inst_16n) {
        g_tls_error = "nb_launch_instances: bad argument";
        return NB_ERR_INVALID;
    }
    NB_LAUNCH_TLS(nbk::launch_instances(count, (const float4 *)pos, (const float4 *)vel, (float4 *)inst_16n, (hipStream_t)stream, overrides().inst_device_libm.on() ? 1u : 0u));
}

NB_EXPORT int nb_launch_cameras(uint32_t count, const void *eyes, const void *dirs, const float *up_xyz, const float *cp16,
                                void *out_16n, void *stream)
{
    if (!count || !eyes || !dirs || !up_xyz || !cp16 || !out_16n) {
        g_tls_error = "nb_launch_cameras: bad argument";
        return NB_ERR_INVALID;
    }
    NB_LAUNCH_TLS(nbk::launch_cameras(count, (const float4 *)eyes, (const float4 *)dirs, up_xyz, cp16, (float4 *)out_16n,
                                      (hipStream_t)stream));
}

NB_EXPORT int nb_launch_random_step(uint32_t first, uint32_t count, void *pos, void *vel, uint64_t seed, uint64_t step, void *stream)
{
    if (!count || !pos || !vel) {
        g_tls_error = "nb_launch_random_step: bad argument";
        return NB_ERR_INVALID;
    }
    NB_LAUNCH_TLS(nbk::launch_random(first, count, (float4 *)pos, (float4 *)vel, seed, step, (hipStream_t)stream));
}

NB_EXPORT int nb_launch_pack(uint32_t count, const void *xyz, void *rec4, void *stream)
{
    if (!count || !xyz || !rec4) {
        g_tls_error = "nb_launch_pack: bad argument";
        return NB_ERR_INVALID;
    }
    NB_LAUNCH_TLS(nbk::launch_pack(count, (const float *)xyz, (float4 *)rec4, (hipStream_t)stream));
}

NB_EXPORT int nb_launch_unpack(uint32_t count, const void *rec4, void *xyz, void *stream)
{
    if (!count || !rec4 || !xyz) {
        g_tls_error = "nb_launch_unpack: bad argument";
        return NB_ERR_INVALID;
    }
    NB_LAUNCH_TLS(nbk::launch_unpack(count, (const float4 *)rec4, (float *)xyz, (hipStream_t)stream));
}

#include "nb_peers.inc"
#include "nb_shard.inc"
