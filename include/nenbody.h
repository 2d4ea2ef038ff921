/*
 * nenbody.h -- C ABI of libnenbody_hip.so: nenbody's all-pairs gravity + Euler step on MI355X (gfx950).
 *
 * This is the drop-in boundary for the per-frame simulation update of Dasch0/nenbody: `update_instance_nbody`
 * (reference src/main.rs:404-441) -- the path this library exists for -- and, as the first row beyond it, the
 * controller the binary actually runs, `update_instance_boids` (src/main.rs:443-526).  Both are bodies of the
 * `Scene::step()` the reference's empty `src/scene.rs` (src/scene.rs:1, declared at src/main.rs:2) was evidently
 * meant to hold.
 * The reference has no FFI of its own; each entry point below cites the reference
 * interface it stands in for.  INTEGRATION.md shows the Rust `extern "C"` block and the
 * `Scene` shim a maintainer would add.  Only what a host binds is declared here; the library's self-tests and the test
 * suite's switches are in nenbody_diag.h.
 *
 * Conventions
 *  - plain pointers and sizes only; no C++/torch types.
 *  - every function returns NB_OK (0) or a negative nb_status; nothing unwinds across the boundary.
 *  - host arrays are caller-allocated, caller-freed, and never retained past the call.
 *  - positions / velocities cross as AoS stride-3 float arrays (x,y,z): the in-memory form of
 *    Vec<cgmath::Point3<f32>> / Vec<cgmath::Vector3<f32>>, so `positions.as_ptr() as *const f32` needs no repack.
 *  - instance matrices cross as 16 floats per body, column-major: the in-memory form of
 *    Vec<[[f32; 4]; 4]> consumed as `mat4 model[]` by shaders/scene.vert:12-14,18.
 *  - a context is single-owner and not re-entrant (the reference calls the update on the winit main
 *    thread, src/main.rs:925); work is queued on a HIP stream and nb_step may return before it finishes.
 *  - there is NO CPU fallback: without a HIP device every compute entry point fails with NB_ERR_NO_DEVICE.
 */
#ifndef NENBODY_H
#define NENBODY_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NB_ABI_VERSION 2

typedef enum nb_status {
    NB_OK = 0,
    NB_ERR_INVALID = -1,     /* bad argument (null pointer, zero bodies, range outside the set, bad mode/tile) */
    NB_ERR_NO_DEVICE = -2,   /* no HIP device visible / HIP runtime failed to initialise */
    NB_ERR_HIP = -3,         /* a HIP call failed; nb_last_error() has the HIP error string */
    NB_ERR_ALLOC = -4,       /* device or host allocation failed */
    NB_ERR_STATE = -5,       /* call sequence error (e.g. nb_step before nb_upload) */
    NB_ERR_UNSUPPORTED = -6  /* e.g. n_devices != 1: one process drives one GPU; see nb_launch_step for sharding */
} nb_status;

/* Arithmetic variants of the pair fold (src/main.rs:425-432). */
typedef enum nb_mode {
    /* Bit-identical to the reference's binary32 arithmetic: sequential j = 0..N-1 per body, no FMA
     * contraction, correctly rounded (vec*G)/dist per component.  The parity path. */
    NB_MODE_STRICT = 0,
    /* Same law, reassociated for speed: r2 by FMA chain, v_rcp_f32 instead of the divide (one reciprocal shared by two
     * pairs where their product cannot leave the normal range), G hoisted out of the sum, j range may be split and
     * partial sums combined in a fixed (deterministic) order.
     * Per-step relative force error ~1e-6; NOT bit-identical to the reference. */
    NB_MODE_FAST = 1
} nb_mode;

/* Constants of the step.  Defaults are the reference's: src/main.rs:411-413. */
typedef struct nb_params {
    float dt;       /* 0.1       main.rs:411 (velocity update only; position advances by v with no dt, main.rs:436) */
    float G;        /* 0.001     main.rs:412 */
    float bias;     /* 0.0000001 main.rs:413, added to the squared distance */
    uint32_t tile;  /* bodies staged through LDS per tile; 0 = library default; else 256, 512 or 1024 */
    uint32_t mode;  /* nb_mode */
} nb_params;

/* Constants of the boids controller, update_instance_boids (src/main.rs:443-526) -- the controller the reference's
 * event loop calls (src/main.rs:925).  Defaults are the reference's: src/main.rs:450-456. */
typedef struct nb_boids_params {
    float dt;               /* 0.04   main.rs:450 */
    float rule_1_distance;  /* 1000.0 main.rs:451; compared with the SQUARED distance (main.rs:474-475) */
    float rule_2_distance;  /* 5.0    main.rs:452 */
    float rule_3_distance;  /* 500.0  main.rs:453; a distance between VELOCITIES (main.rs:497) */
    float rule_1_scale;     /* 0.02   main.rs:454 */
    float rule_2_scale;     /* 0.05   main.rs:455 */
    float rule_3_scale;     /* 0.5    main.rs:456 */
    uint32_t tile;          /* bodies staged through LDS per tile; 0 = library default; else 256, 512 or 1024 */
} nb_boids_params;

typedef struct nb_ctx nb_ctx; /* opaque; owns the device buffers and the stream */

/* -- library ------------------------------------------------------------------------------------------- */
int nb_abi_version(void);
/* Fills *p with the reference constants (main.rs:411-413), tile 0, NB_MODE_STRICT. */
void nb_default_params(nb_params *p);
/* Number of HIP devices this process can see, or a negative nb_status. */
int nb_device_count(void);
/* Message of the last failure on `ctx`, or (ctx == NULL) of the calling thread's last failed
 * context-free call.  Never NULL; valid until the next call on the same ctx / thread. */
const char *nb_last_error(const nb_ctx *ctx);

/* Seeded stand-in for the reference's unseeded rand::thread_rng() initial state (src/main.rs:736-747):
 * same distributions and draw order (all velocities (U[0,0.1),U[0,0.1),0) first, then all positions
 * (U[-100,100),U[-100,100),0)).  Host-side; pos_xyz / vel_xyz hold 3*n floats. */
int nb_init_state(uint64_t seed, uint32_t n, float *pos_xyz, float *vel_xyz);

/* -- context API: what a Scene owns -------------------------------------------------------------------- *
 * Stands in for the state the reference keeps in main(): positions, velocities, old_positions,
 * old_velocities, instance_data (src/main.rs:738-750) -- here device-resident.                            */

/* Create a context for n bodies on the current HIP device.  n_devices must be 1 (one process per GPU;
 * multi-GPU hosts shard with nb_launch_step + an all-gather, see below).  params == NULL -> defaults. */
int nb_create(uint32_t n, uint32_t n_devices, const nb_params *params, nb_ctx **out);
void nb_destroy(nb_ctx *ctx);

/* Host -> device.  pos_xyz, vel_xyz: 3*n floats each (the Vec<Point3>/Vec<Vector3> the reference's
 * update function receives, src/main.rs:406-408). */
int nb_upload(nb_ctx *ctx, const float *pos_xyz, const float *vel_xyz);

/* k applications of update_instance_nbody (src/main.rs:404-441), device-resident, asynchronous.
 * old_positions (main.rs:415) is the ping-pong buffer; old_velocities (main.rs:416) is never read by
 * the reference's function and has no counterpart. */
int nb_step(nb_ctx *ctx, uint32_t k);

/* Fills *p with the reference constants (main.rs:450-456), tile 0. */
void nb_boids_default_params(nb_boids_params *p);

/* k applications of update_instance_boids (src/main.rs:443-526) to the context's state, device-resident,
 * asynchronous; bit-identical to the reference's binary32 arithmetic (the three folds run in index order, the
 * sqrt-based radius tests are evaluated exactly).  params == NULL -> defaults.  May be mixed freely with nb_step. */
int nb_step_boids(nb_ctx *ctx, uint32_t k, const nb_boids_params *params);

/* k applications of the random-walk controller update_instance_random (src/main.rs:381-402): vel += (U[-1e-4,1e-4),
 * U[-1e-4,1e-4), 0); pos += vel.  The reference draws from an unseeded thread_rng, so only the distribution is kept:
 * the stream here is counter based on (seed, step index since nb_upload, body index). */
int nb_step_random(nb_ctx *ctx, uint32_t k, uint64_t seed);

/* Device-resident hand-off: pointers to the context's CURRENT device buffers -- 16-byte position
 * and velocity records and, when inst is non-NULL, the model matrices of the current state (16 floats per body,
 * produced now on the context's stream) -- for a consumer that reads them on the GPU instead of through nb_download
 * (the reference re-uploads instance_data every frame, src/main.rs:932-936).  Valid until the next nb_step*, nb_upload
 * or nb_destroy on ctx; call nb_sync first if the consumer does not run on a stream ordered after ctx's. */
int nb_device_state(nb_ctx *ctx, const void **pos_rec, const void **vel_rec, const void **inst_16n);

/* CameraArray::update (src/gfx.rs:397-408, build_camera :358-369) for the context's current state: per body
 * out = cp * look_at_dir(eye = position, dir = velocity, up), where cp16 = OPENGL_TO_WGPU_MATRIX * perspective(...) is the
 * array's constant (16 floats, column-major; nb_camera_constant below, or the caller's own cgmath::perspective).  out_16n: host, 16n floats. */
int nb_cameras(nb_ctx *ctx, const float *up_xyz, const float *cp16, float *out_16n);

/* That constant: cp16 = OPENGL_TO_WGPU_MATRIX * cgmath::perspective(Deg(vertical_fov_deg), aspect_ratio, near, far), 16 floats,
 * column-major (src/gfx.rs:12-17, 365, 367; the reference passes near = 1, far = 10000 and vertical_fov = horizontal_fov /
 * aspect_ratio, src/gfx.rs:381).  Host arithmetic, no device needed; NB_ERR_INVALID where cgmath's assertions would panic. */
int nb_camera_constant(float vertical_fov_deg, float aspect_ratio, float near_plane, float far_plane, float *cp16);

/* Device -> host, after waiting for queued steps.  Any of the three may be NULL.
 * inst_16n, when given, receives the model matrices of the current state (src/main.rs:437-439),
 * produced on demand by a separate kernel: they never feed back into the dynamics. */
int nb_download(nb_ctx *ctx, float *pos_xyz, float *vel_xyz, float *inst_16n);

/* Block until all queued work of ctx has finished. */
int nb_sync(nb_ctx *ctx);

/* Steps taken since nb_upload. */
uint64_t nb_steps_done(const nb_ctx *ctx);

/* -- one-call drop-ins: the reference's free functions themselves ------------------------------------- *
 * `update_instance_nbody(instances, positions, old_positions, velocities, old_velocities)` (src/main.rs:404-410) and
 * `update_instance_boids` (src/main.rs:443-449) with their own five slices, updated in place; every length is the
 * slice's len() in elements (bodies).  A maintainer replaces each function body by one call.
 * Kept from the reference:
 *   - old_positions / old_velocities receive copies of positions / velocities first (main.rs:415-416, 459-460);
 *     unequal lengths are NB_ERR_INVALID, where copy_from_slice panics;
 *   - instances.zip(positions).zip(velocities) stops at the shortest (main.rs:420-423, 465-469): only that many
 *     bodies are written, while the folds still run over all of old_positions;
 *   - boids folds rule 1 and rule 2 over old_positions.iter() (main.rs:471, 482) and rule 3 over old_velocities.iter()
 *     (main.rs:494): each fold has its own slice's length, nothing is indexed by the other's, so positions and
 *     velocities of different lengths are served as the reference serves them (the zip bounds which bodies move).
 * One upload, one step, one download per call.  The device contexts live inside the library between calls: up to three,
 * keyed by controller, body count and constants (a host may alternate controllers or entity counts from frame to frame; only
 * a fourth shape rebuilds one); calls are serialised by an internal lock.
 * The call returns when the results are in the caller's arrays.  Up to 2 048 bodies (the reference's ceiling, main.rs:653) the
 * calling thread SPINS on a word the last kernel writes behind the results (4 us sooner than a stream wait; at most 2 ms, then it
 * falls back to the stream wait); larger sets block in hipStreamSynchronize.
 * params == NULL -> the reference constants. */
int nb_update_instance_nbody(float *instances_16n, size_t n_instances, float *positions_xyz, size_t n_positions,
                             float *old_positions_xyz, size_t n_old_positions, float *velocities_xyz, size_t n_velocities,
                             float *old_velocities_xyz, size_t n_old_velocities, const nb_params *params);
int nb_update_instance_boids(float *instances_16n, size_t n_instances, float *positions_xyz, size_t n_positions,
                             float *old_positions_xyz, size_t n_old_positions, float *velocities_xyz, size_t n_velocities,
                             float *old_velocities_xyz, size_t n_old_velocities, const nb_boids_params *params);
/* update_instance_random(instances, positions, velocities) (src/main.rs:381-385) as one call: the reference's three slices,
 * the first min(n_instances, n_positions, n_velocities) bodies move (its zip, :386-389), the rest are not touched.  The
 * reference draws from an unseeded thread_rng; here body n at the library's k-th call draws from the counter-based stream
 * (seed, k, n) of nb_step_random, with the seed and the call counter kept inside the library (nb_update_random_seed sets the
 * seed and restarts the counter; the default seed is fixed, so a run is reproducible): the distribution is the reference's,
 * the draws are not.  nb_update_instance_random_seeded is the same step with the stream position given by the caller
 * (independent of how a run is split across calls or hosts). */
int nb_update_instance_random(float *instances_16n, size_t n_instances, float *positions_xyz, size_t n_positions,
                              float *velocities_xyz, size_t n_velocities);
int nb_update_instance_random_seeded(float *instances_16n, size_t n_instances, float *positions_xyz, size_t n_positions,
                                     float *velocities_xyz, size_t n_velocities, uint64_t seed, uint64_t step);
void nb_update_random_seed(uint64_t seed);

/* Frees the contexts the calls above keep (otherwise they are reclaimed with the process). */
void nb_update_release(void);

/* -- launch API: caller-owned device memory ------------------------------------------------------------ *
 * For hosts that own the device buffers and the exchange step themselves (one process per GPU, RCCL
 * all-gather of positions between steps).  Device layout: one 16-byte record per body,
 * float4 (x, y, z, 0) for positions and (vx, vy, vz, 0) for velocities.
 * `stream` is a hipStream_t (NULL = the default stream).  All launches are asynchronous.                 */

/* Bytes of device scratch nb_launch_step needs for this shape (may be 0): the scalar-load kernels keep x / y / z planes of the
 * position set there (12 B per body of the WHOLE set), FAST its rows of partial sums as well (a whole set in the pairs form:
 * n_total x n_total / 2048 x 12 B -- 100 MB at 131 072 bodies).  Always ask; the number changes with shape and mode. */
size_t nb_scratch_bytes(const nb_params *params, uint32_t n_total, uint32_t count);

/* One step for bodies [first, first+count) of a set of n_total:
 *   pos_in   n_total records, the start-of-step snapshot (old_positions, main.rs:415), read only
 *   pos_out  n_total records; only [first, first+count) is written (the caller gathers the rest)
 *   vel      count records, this shard's velocities, updated in place (local index i - first)
 * pos_out must not alias pos_in. */
int nb_launch_step(const nb_params *params, uint32_t n_total, uint32_t first, uint32_t count, const void *pos_in,
                   void *pos_out, void *vel, void *scratch, size_t scratch_bytes, void *stream);

/* A FAST step in two phases, so that the exchange of a multi-GPU job can overlap with compute (SURVEY.md section 8e,
 * "Overlap"): NB_PHASE_RANGE folds records [j_lo, j_hi) of pos_in only -- e.g. this rank's own slot of the snapshot, which
 * is in place before the all-gather of the other slots has landed -- into `scratch`; NB_PHASE_REST folds the rest of the
 * set, adds every partial sum in a fixed order and integrates (main.rs:434-436).  Both calls take the same arguments; the
 * order of the additions differs from a one-call step (legal in FAST only: STRICT keeps the reference's j = 0..N-1 order
 * and returns NB_ERR_UNSUPPORTED here).  scratch: nb_scratch_bytes_phased() bytes, untouched between the two calls. */
enum { NB_PHASE_RANGE = 0, NB_PHASE_REST = 1 };
size_t nb_scratch_bytes_phased(const nb_params *params, uint32_t n_total, uint32_t count, uint32_t j_lo, uint32_t j_hi);
int nb_launch_step_phase(const nb_params *params, uint32_t n_total, uint32_t first, uint32_t count, uint32_t j_lo, uint32_t j_hi,
                         int phase, const void *pos_in, void *pos_out, void *vel, void *scratch, size_t scratch_bytes, void *stream);

/* FAST on shards with every UNORDERED pair evaluated once ("half shell").  The term of the pair (a, b) in a's sum is the negative
 * of its term in b's (src/main.rs:428-430), so one evaluation serves both bodies even when they live on different GPUs, provided
 * the other body's half is sent to its owner: a step then has TWO exchanges instead of one, and half the arithmetic.
 * Rank r of P = n_total / count equal ranks (first = r * count) evaluates the pairs of its bodies with the N/2 bodies that follow
 * them on the ring of indices; with D = nb_ring_partners() the ranks r+1 .. r+D (mod P) own those bodies.
 *   nb_launch_ring_fold    pos_in: n_total records (the snapshot, as for nb_launch_step).  sums: (D + 1) * count records
 *                          (x, y, z, 0), written: [0, count) = this rank's own bodies, [d * count, (d + 1) * count) = the
 *                          sums this rank evaluated for the bodies of rank (r + d) mod P.
 *   -- the host's second exchange: chunk d of `sums` goes to rank (r + d) mod P, which stores it as chunk d - 1 of its `recv`
 *      (D * count records; chunk d - 1 comes from rank (r - d) mod P), d = 1..D --
 *   nb_launch_ring_finish  adds sums[0, count) and the D received chunks in ascending distance, then src/main.rs:434-436:
 *                          pos_out[first, first + count) and vel (count records) as nb_launch_step writes them.
 * Every order of addition is fixed (run-to-run identical); FAST only (reassociated sums; STRICT shards keep nb_launch_step).
 * nb_ring_partners: D >= 1; 0 when this shape does not take the form (STRICT, unequal or partial ranks, count not a multiple of
 * 256, or fewer than 2^30 ordered pairs per rank and step -- n_total x count -- where what the form saves no longer pays for its
 * second exchange: use nb_launch_step); or a negative nb_status.  Every rank of a job gets the same answer. */
int nb_ring_partners(const nb_params *params, uint32_t n_total, uint32_t first, uint32_t count);
size_t nb_ring_scratch_bytes(const nb_params *params, uint32_t n_total, uint32_t first, uint32_t count);
int nb_launch_ring_fold(const nb_params *params, uint32_t n_total, uint32_t first, uint32_t count, const void *pos_in, void *sums,
                        void *scratch, size_t scratch_bytes, void *stream);
int nb_launch_ring_finish(const nb_params *params, uint32_t n_total, uint32_t first, uint32_t count, const void *pos_in, void *pos_out,
                          void *vel, const void *sums, const void *recv, void *stream);

/* The same step in PHASES, so that its exchanges can hide behind compute (round 5; SURVEY.md section 8e "Overlap";
 * src/main.rs:415, 425-432: every pair reads the start-of-step snapshot only).  A rank's pairs fall into those whose two bodies
 * it owns itself -- they need nothing from another GPU -- and those with the bodies of the D ranks in front:
 *   NB_RING_OWN    pairs inside the rank's own slot, as many as one round of workgroups holds (3 068 of the 4 224 own-slot
 *                  sub-tiles of an 8-rank share of 131 072 bodies: 23 us on an MI355X).  Reads ONLY records
 *                  [first, first + count) of pos_in -- which the rank's own nb_launch_ring_finish wrote -- so it may run while
 *                  the all-gather of the other slots of pos_in is still landing.
 *   NB_RING_REST   every other pair (the whole snapshot must be in place), and the sums of the halves that belong to the ranks
 *                  in front: records [count, (D + 1) count) of `sums` are final when it completes -- the second exchange can start.
 *   NB_RING_SUMS   the rank's own sums, records [0, count) of `sums`: it can run beside the second exchange.
 *   then nb_launch_ring_finish as above.
 * One step, with `x` a second stream for the exchanges:
 *     OWN | wait(all-gather of the last step) | REST | [x: second exchange] SUMS | wait(x) | finish | [x: all-gather]
 * (a host whose cross-stream waits are dear keeps the second exchange on the compute stream -- REST | SUMS | exchange | finish --
 * as nb_shard_step does: two waits cost more than the 9 us of SUMS they would hide; the all-gather's two waits hide behind OWN)
 * Same arguments in every call of a step (same scratch, untouched between them: nb_ring_scratch_bytes() covers both forms).
 * The sums are added in another (fixed) order than nb_launch_ring_fold's: run-to-run identical, FAST's tolerances, its own bits.
 * nb_ring_phased: 1 where the shape can run its step this way (nb_ring_partners() > 0 and the rank's rows fit one launch -- every
 * rank count of BASELINE's config 4; config 5's ranks walk their rows in groups and keep nb_launch_ring_fold: two exchanges of
 * 2 MB per peer against 13.6 ms of compute), 0 where it keeps nb_launch_ring_fold; or a negative nb_status. */
enum { NB_RING_OWN = 1, NB_RING_REST = 2, NB_RING_SUMS = 3, NB_RING_OWN_READY = 4 };
int nb_ring_phased(const nb_params *params, uint32_t n_total, uint32_t first, uint32_t count);
int nb_launch_ring_fold_phase(const nb_params *params, uint32_t n_total, uint32_t first, uint32_t count, int phase, const void *pos_in,
                              void *sums, void *scratch, size_t scratch_bytes, void *stream);
/* The finish of a step in phases, FUSED (two launches fewer per step: every small launch of a step costs ~5 us of ramp and drain):
 *   nb_launch_ring_finish_phase  with sums == NULL adds the rank's own records ITSELF -- NB_RING_SUMS is not launched; the same
 *                                additions in the same order, the same bits -- (a host that runs NB_RING_SUMS beside its second
 *                                exchange passes `sums`: records [0, count) are read instead); then the D received chunks and
 *                                src/main.rs:434-436 as nb_launch_ring_finish; and leaves the planes and flag words of the NEW own
 *                                slot (pos_out[first, first + count)) in `scratch`.
 *   NB_RING_OWN_READY            NB_RING_OWN of the NEXT step without the launch that prepares its planes: valid when the last thing
 *                                that touched `scratch` was nb_launch_ring_finish_phase of the same shape and this call's pos_in is
 *                                that call's pos_out.  (pos_in is not read.)
 *     step k:  OWN_READY | wait(all-gather k - 1) | REST | second exchange | finish_phase | [all-gather k] ...
 * The first step of a run, and any step behind something else that used `scratch`, starts with NB_RING_OWN. */
int nb_launch_ring_finish_phase(const nb_params *params, uint32_t n_total, uint32_t first, uint32_t count, const void *pos_in, void *pos_out,
                                void *vel, const void *sums, const void *recv, void *scratch, size_t scratch_bytes, void *stream);

/* One boids step (main.rs:443-526) for bodies [first, first+count) of a set of n_total:
 *   pos_in, vel_in    n_total records each: the snapshots old_positions / old_velocities (main.rs:459-460), read only
 *   pos_out, vel_out  n_total records each; only [first, first+count) is written (a multi-GPU caller all-gathers BOTH)
 * The outputs must not alias the inputs. */
int nb_launch_boids_step(const nb_boids_params *params, uint32_t n_total, uint32_t first, uint32_t count, const void *pos_in,
                         const void *vel_in, void *pos_out, void *vel_out, void *stream);

/* The same step with the j range cut into slices ("split" form, for shards whose bodies alone cannot fill the chip: a rank of an
 * 8-GPU job): every radius test is evaluated on the reference's operands with the reference's bits, so the neighbour SETS and
 * COUNTS are the reference's, but a body's sums are added slice by slice (in index order inside a slice, then in slice order --
 * the same order every step) instead of in one index-ordered chain: not bit-identical -- as close to the exact sums as the
 * reference's own sequential binary32 sums are (the tests hold it to that, against the same sums carried in binary64), which
 * puts it within ~1e-5 of the bit-exact step's velocities per step on tens of thousands of bodies.  Opt-in: nb_launch_boids_step stays the reference's arithmetic.  scratch: nb_boids_split_scratch_bytes(). */
size_t nb_boids_split_scratch_bytes(const nb_boids_params *params, uint32_t n_total, uint32_t count);
int nb_launch_boids_step_split(const nb_boids_params *params, uint32_t n_total, uint32_t first, uint32_t count, const void *pos_in,
                               const void *vel_in, void *pos_out, void *vel_out, void *scratch, size_t scratch_bytes, void *stream);

/* Model matrices (main.rs:437-439) for `count` bodies: pos, vel -> inst (16 floats per body). */
int nb_launch_instances(uint32_t count, const void *pos, const void *vel, void *inst_16n, void *stream);

/* Camera matrices (gfx.rs:397-408) for `count` entities on caller-owned device memory: eyes/dirs are 16-byte records,
 * out receives 16 floats per entity. */
int nb_launch_cameras(uint32_t count, const void *eyes, const void *dirs, const float *up_xyz, const float *cp16, void *out_16n,
                      void *stream);

/* One random-walk step (main.rs:381-402) in place for `count` bodies whose global indices start at `first`. */
int nb_launch_random_step(uint32_t first, uint32_t count, void *pos, void *vel, uint64_t seed, uint64_t step, void *stream);

/* Waits for `stream` and reports a failure a launch-API kernel recorded on the current device since the last call:
 * NB_ERR_STATE when a STRICT block-chain workgroup gave up waiting for its turn (its outputs were written as NaN), NB_OK
 * otherwise.  The context and shard objects make the same check themselves in nb_sync / nb_download / nb_shard_sync /
 * nb_shard_download; a host that drives nb_launch_step directly calls this wherever it waits for the device. */
int nb_launch_status(void *stream);

/* Stride-3 <-> 16-byte record conversion on the device. */
int nb_launch_pack(uint32_t count, const void *xyz, void *rec4, void *stream);
int nb_launch_unpack(uint32_t count, const void *rec4, void *xyz, void *stream);

/* -- the exchanges of a multi-GPU step as PULLS over xGMI (round 5) ---------------------------------------- *
 * No collective library, no second stream, no kernel that polls: every rank registers a few device buffers (the two position
 * replicas, the pairs form's `sums`) and hands the others a blob of IPC memory handles (nb_peers_export; the host moves the blobs:
 * MPI, torch.distributed, a file); after nb_peers_import every rank holds mappings of every other rank's buffers and of a block of
 * flag words.  An exchange is then, on the rank's OWN stream:
 *     nb_peers_signal   hipStreamWriteValue32(my flag[channel] = next count): "what I launched so far is complete"
 *     ... whatever needs no other rank (the pairs inside the rank's own slot: nb_launch_ring_fold_phase(NB_RING_OWN)) ...
 *     nb_peers_gather   hipStreamWaitValue32(peer's flag[channel] >= count) for every peer, then ONE kernel copies slot q of peer q's
 *                       buffer into slot q of this rank's (16-byte loads over xGMI)
 *     nb_peers_ring     the same for the pairs form's second exchange: chunk d of rank (r - d)'s buffer -> chunk d - 1 of `recv`
 * Nobody writes another rank's memory, and a kernel reads a peer's memory only behind that peer's signal.  Buffers may be pieces of
 * larger allocations (a caching allocator's).  Up to 16 ranks (one xGMI domain), all in processes of one node.  Ranks call the
 * same sequence of signals and exchanges per channel (SPMD).  That a buffer is not rewritten while a peer still pulls from it is the
 * CALLER's ordering: alternate the two position replicas (as every host here does) and rewrite `sums` only behind a gather.
 * nb_shard_use_peers / ShardedScene(exchange="peers") wire this into a step.  Measured between two processes on one GPU: a signal +
 * wait + 256 KB pull round trip in ~10 us (profiles/r05/ubench_ipc.log); between GPUs: not run in this build -- verify_exchanges
 * checks a pattern through it before the first step, and falls back. */
#define NB_PEERS_MAX_BUFFERS 4
#define NB_PEERS_CHANNELS 4
typedef struct nb_peers nb_peers;
size_t nb_peers_blob_bytes(void);
int nb_peers_create(int rank, int world, nb_peers **out);
void nb_peers_destroy(nb_peers *p);   /* (after a probe that had to abandon its stream: releases nothing -- every release would wait for that stream) */
const char *nb_peers_last_error(const nb_peers *p);
int nb_peers_export(nb_peers *p, void *const *bufs, const size_t *bytes, int nbufs, void *blob);
int nb_peers_import(nb_peers *p, const void *blobs /* world x nb_peers_blob_bytes(), rank-major */);
/* First contact with a deadline (collective), in two stages: this rank signals and the HOST reads every peer's word through the mapping
 * until it shows the signal or timeout_ms has passed (nothing can block); only then a wait / one-record pull round on a stream of its
 * own, watched for the same time (a wait the command processor cannot satisfy is released by hand).  NB_ERR_STATE: keep the collectives.
 * Hosts call it once after nb_peers_import (ShardedScene.setup_peers and nb_shard_peer_import do). */
int nb_peers_probe(nb_peers *p, uint32_t timeout_ms);
int nb_peers_signal(nb_peers *p, int channel, void *stream);
int nb_peers_gather(nb_peers *p, int channel, int buf, size_t slot_bytes, void *stream);
int nb_peers_ring(nb_peers *p, int channel, int buf, void *recv, size_t chunk_bytes, int partners, void *stream);

/* -- sharded scene: one process (or thread) per GPU ------------------------------------------------------ *
 * The host side of the multi-GPU path in the library itself, for hosts that do not bring torch: rank r of `world`
 * owns bodies [r*slot, r*slot + count) with slot = ceil(n / world) (trailing ranks may be short or empty) and a replica
 * of all positions; one step is the local update followed by ONE exchange in which every rank contributes its slot of
 * the new positions and receives the others' (`old_positions` of src/main.rs:415 is the replica).
 * The boids controller exchanges velocities the same way (it reads every old velocity, src/main.rs:494-504).
 * STRICT results do not depend on `world`.  The exchange is either RCCL (ncclAllGather in place, loaded with dlopen:
 * NENBODY_RCCL, else the librccl.so.1 already in the process, else the system's) or a function the host supplies. */
typedef struct nb_shard nb_shard;

/* The exchange: `buf` is device memory of world*slot_bytes; this rank's slot (offset rank*slot_bytes) is ready on
 * `stream` (a hipStream_t); on return (or: ordered on `stream`) every other slot must hold that rank's contribution.
 * Return 0, or nonzero to fail the step with NB_ERR_STATE. */
typedef int (*nb_gather_fn)(void *user, void *buf, size_t slot_bytes, int rank, int world, void *stream);

#define NB_COMM_ID_BYTES 128
/* A fresh RCCL unique id (ncclGetUniqueId): call on one rank, hand the NB_COMM_ID_BYTES to the others by any channel. */
int nb_comm_id(void *id);

/* Rank `rank` of `world` for n bodies on the current HIP device.  params == NULL -> defaults. */
int nb_shard_create(uint32_t n, int rank, int world, const nb_params *params, nb_shard **out);
void nb_shard_destroy(nb_shard *sh);
/* Choose the exchange (required before stepping when world > 1).  nb_shard_use_rccl is collective over the ranks
 * (ncclCommInitRank with the id from nb_comm_id). */
int nb_shard_use_rccl(nb_shard *sh, const void *id);
int nb_shard_use_gather(nb_shard *sh, nb_gather_fn fn, void *user);
/* FAST with equal ranks of whole blocks takes the pairs form on shards (nb_launch_ring_fold above: every unordered pair once, a
 * second exchange per step) where the second exchange exists: with RCCL it is a group of ncclSend / ncclRecv; a host that brings
 * its own all-gather (nb_shard_use_gather) brings this one too, or keeps the ordered fold.  `send`: `partners` chunks of
 * chunk_bytes, chunk d - 1 for rank (rank + d) % world; `recv`: chunk d - 1 from rank (rank - d) % world; both device memory,
 * `send` ready on `stream`, `recv` must be complete or ordered on `stream` on return.  nb_shard_set_pairs(sh, 0) keeps the
 * ordered fold whatever the shape; nb_shard_pairs_partners: the D a step will use (0: the ordered fold). */
/* Both exchanges as PULLS over xGMI (nb_peers_* above; round 5): nb_shard_peer_export fills this rank's blob of nb_peers_blob_bytes()
 * bytes, the host hands every rank ALL blobs, rank-major (any channel), nb_shard_peer_import maps the others' position replicas and
 * `sums`.  From then on the all-gather and the pairs form's second exchange are a signal, stream waits on the peers' flag words and
 * one copy kernel, all on the shard's own stream; with nb_shard_set_overlap the signal goes out behind the finish kernel and the waits
 * come behind the next step's own-slot pairs -- no second stream, no idle gap.  No RCCL communicator is needed (one may exist beside it:
 * the boids controller, which moves another buffer, uses the exchange chosen before).  All ranks in processes of ONE node, at most 16.
 * nb_shard_use_peers(sh, 0 / 1) switches between this and the exchange chosen before.  Collective in effect: every rank imports
 * before any rank steps.  nb_shard_verify_exchanges checks it on a pattern like any other exchange. */
int nb_shard_peer_export(nb_shard *sh, void *blob);
int nb_shard_peer_import(nb_shard *sh, const void *blobs);
int nb_shard_use_peers(nb_shard *sh, int on);
typedef int (*nb_ring_fn)(void *user, const void *send, void *recv, size_t chunk_bytes, int partners, int rank, int world, void *stream);
int nb_shard_use_ring(nb_shard *sh, nb_ring_fn fn, void *user);
int nb_shard_set_pairs(nb_shard *sh, int on);
int nb_shard_pairs_partners(const nb_shard *sh);
/* on != 0: nb_shard_step_boids takes the split form of the boids step (nb_launch_boids_step_split: the reference's neighbour sets
 * and counts, sums reassociated) -- what lets eight ranks run the controller about eight times as fast as one instead of 3.4
 * times; off (the default): bit-identical to the reference whatever the world size. */
int nb_shard_set_boids_split(nb_shard *sh, int on);
/* FAST only ("Overlap"): with `on` != 0 the exchanges of a step run on a second stream, behind compute that
 * does not need them.  A shard in the ordered fold folds its own slot of the snapshot while the all-gather of the other slots is
 * still in flight, waits for it, then folds the rest (the two phases of nb_launch_step_phase).  A shard in the pairs form
 * (nb_shard_pairs_partners() > 0, nb_ring_phased()) runs the phases of nb_launch_ring_fold_phase: pairs inside its own slot while
 * the all-gather lands on the second stream, every other pair, the sums, the second exchange (on the shard's own stream), finish
 * (round 5).  The order of the
 * additions changes, which FAST may and STRICT may not: a STRICT shard (and a world of one) ignores the request and stays
 * kernel -> exchange in sequence.  Host-supplied exchanges (nb_shard_use_gather / nb_shard_use_ring) receive the second stream
 * and must order their work on it. */
int nb_shard_set_overlap(nb_shard *sh, int on);
/* Both exchanges of a step, ONCE, on a known per-rank pattern, checked on every rank (collective; the ranks agree on every verdict
 * through a third collective: ncclAllReduce, or the host's gather).  A mismatch moves the exchange to its fallback without restarting
 * anything: the all-gather from a copy of the slot (*gather_path 1; 0: in place), the second exchange as one group per distance
 * (*ring_path 1; 0: one group), then not at all (*ring_path 2: the ordered fold and its one exchange); *ring_path -1: no pairs form
 * planned.  A shard that pulls its exchanges over xGMI (nb_shard_peer_import) is checked as such first (*gather_path 2, *ring_path 3)
 * -- TWICE from the same buffers with different patterns: a reader served from lines it cached at the first pull shows the first pattern
 * again -- and goes back to the exchange chosen before on a mismatch.  Either pointer may be NULL.  A world of one answers at once.  With RCCL on more than one rank the first nb_shard_step runs
 * this by itself if the host has not. */
int nb_shard_verify_exchanges(nb_shard *sh, int *gather_path, int *ring_path);
/* Which form should a FAST step take on THIS machine?  Times `steps` steps (0: four) of every form the shard can take -- 1: the pairs
 * form with its two exchanges in sequence, 2: the same in phases with the exchanges behind compute (nb_shard_set_overlap), 0: the
 * ordered fold with its one exchange -- on the state in hand, takes the slowest rank's time of each (collective), keeps the fastest
 * for the steps to come (*chosen) and puts the state back; ms3 (NULL or three doubles): milliseconds per step of forms 0, 1, 2
 * (negative: not offered).  The library's own line for the pairs form (nb_ring_partners) was drawn from one-GPU timings; this asks
 * the machine.  With RCCL on more than one rank the first nb_shard_step asks by itself unless the host has decided
 * (nb_shard_set_pairs) or asked (this call).  Note: which form runs decides the rounding of FAST's sums -- a host that needs
 * run-to-run identical bits names the form. */
int nb_shard_choose_form(nb_shard *sh, uint32_t steps, int *chosen, double *ms3);
/* 1 where a step of this shard takes the pairs form in phases with both exchanges on the second stream, else 0. */
int nb_shard_pairs_overlapped(const nb_shard *sh);
/* This rank's index range. */
int nb_shard_range(const nb_shard *sh, uint32_t *first, uint32_t *count);
/* Host -> device: ALL n positions and ALL n velocities (identical on every rank; the rank keeps its own velocities). */
int nb_shard_upload(nb_shard *sh, const float *pos_xyz, const float *vel_xyz);
/* k steps of update_instance_nbody / update_instance_boids, asynchronous; collective over the ranks. */
int nb_shard_step(nb_shard *sh, uint32_t k);
int nb_shard_step_boids(nb_shard *sh, uint32_t k, const nb_boids_params *params);
/* Device -> host: all n positions (the replica), this rank's `count` velocities and model matrices.  Any may be NULL. */
int nb_shard_download(nb_shard *sh, float *pos_xyz, float *vel_xyz_local, float *inst_16n_local);
int nb_shard_sync(nb_shard *sh);
/* Message of the last failure on `sh` (never NULL); creation failures are reported through nb_last_error(NULL). */
const char *nb_shard_last_error(const nb_shard *sh);

#ifdef __cplusplus
}
#endif
#endif /* NENBODY_H */
