/*
 * nbody_oracle.c -- CPU restatement of nenbody's all-pairs gravity + Euler step.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check in
 * __graft_entry__.py and bench.py's cpu_baseline leg may load it.  The shipped
 * library (nenbody_amd/lib/libnenbody_hip.so) never links, loads or calls it.
 *
 * PARITY UNPINNED: the reference (Rust) cannot be compiled in this environment
 * (no cargo/rustc) and ships no tests, golden vectors or fixtures for this path
 * (SURVEY.md section 4 / 8c).  This file restates src/main.rs:404-441 operation by
 * operation; it is cross-checked by an independent numpy float32 restatement, by
 * hand-derived known-answer tests (tests/test_oracle.py) and -- what "every operation
 * rounded once to binary32, in the reference's order" MEANS, independently of this
 * compiler and FPU -- by the same steps in exact rational arithmetic with the rounding
 * done by hand (tests/test_oracle_exact.py: n-body, boids, cameras).  None of that is the
 * Rust original's own output: parity with the reference stays unpinned.
 *
 * The arithmetic lives in the third-party crate cgmath 0.17.0 (Cargo.toml:16,
 * Cargo.lock:177-185), which is not vendored under /root/reference.  Its published
 * semantics, as used at the reference's call sites:
 *   Point3 - Point3, Vector3 +/- Vector3    component-wise
 *   Vector3 * f32, Vector3 / f32            component-wise (a true IEEE divide per component)
 *   MetricSpace::distance2(self, other)     (other - self).magnitude2() = dot(d, d)
 *   dot(a, b) for Vector3                   (a.x*b.x + a.y*b.y) + a.z*b.z   (sum of products, left to right)
 *   Matrix4::from_translation(v)            identity with column 3 = (v.x, v.y, v.z, 1)
 *   Matrix4::from_angle_z(t)                (s, c) = sin_cos(t); columns (c,s,0,0), (-s,c,0,0), (0,0,1,0), (0,0,0,1)
 *   Matrix4 * Matrix4                       column k = ((a*r[k][0] + b*r[k][1]) + c*r[k][2]) + d*r[k][3],
 *                                           a..d = columns of lhs, Vector4 ops component-wise
 *   Angle::atan2(y, x)                      Rad(y.atan2(x)) -> libm atan2f
 * rustc/LLVM neither contracts a*b+c into an FMA nor reassociates float adds without
 * fast-math, so this file MUST be built with -ffp-contract=off and without
 * -ffast-math (see oracle/Makefile).  All arithmetic is IEEE binary32.
 *
 * Memory layout at this boundary: positions and velocities are AoS stride-3 float
 * arrays (x,y,z), i.e. the in-memory form of Vec<Point3<f32>> / Vec<Vector3<f32>>;
 * instance matrices are 16 floats per body, column-major ([[f32;4];4]).
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NBO_API __attribute__((visibility("default")))

/* src/main.rs:411-413 */
#define NBO_DT 0.1f
#define NBO_G 0.001f
#define NBO_BIAS 0.0000001f

NBO_API int nbo_abi_version(void) { return 1; }

NBO_API void nbo_default_constants(float *dt, float *G, float *bias)
{
    *dt = NBO_DT;
    *G = NBO_G;
    *bias = NBO_BIAS;
}

/* ------------------------------------------------------------------------------------
 * Seeded initial conditions.  The reference draws from rand::thread_rng(), which is
 * unseeded and cannot be reproduced (src/main.rs:737).  This generator is build-owned;
 * it keeps the reference's DISTRIBUTIONS and DRAW ORDER (src/main.rs:738-747):
 *   first, for every body: velocity = (U[0,0.1), U[0,0.1), 0)
 *   then,  for every body: position = (U[-100,100), U[-100,100), 0)
 * Bit source: splitmix64; uniform: 24 high bits -> u in [0,1), value = lo + (hi-lo)*u in
 * binary32 (one multiply, one add).  nenbody_amd's nb_init_state must produce the same bits.
 * ---------------------------------------------------------------------------------- */
static uint64_t splitmix64(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static float uniform_f32(uint64_t *s, float lo, float hi)
{
    uint32_t bits24 = (uint32_t)(splitmix64(s) >> 40);
    float u = (float)bits24 * 0x1.0p-24f;
    float scale = hi - lo;
    float v = scale * u;
    return lo + v;
}

NBO_API void nbo_init_state(uint64_t seed, uint32_t n, float *pos3, float *vel3)
{
    uint64_t s = seed;
    for (uint32_t i = 0; i < n; ++i) { /* src/main.rs:738-742 */
        vel3[3 * i + 0] = uniform_f32(&s, -0.0f, 0.1f);
        vel3[3 * i + 1] = uniform_f32(&s, -0.0f, 0.1f);
        vel3[3 * i + 2] = 0.0f;
    }
    for (uint32_t i = 0; i < n; ++i) { /* src/main.rs:743-747 */
        pos3[3 * i + 0] = uniform_f32(&s, -100.0f, 100.0f);
        pos3[3 * i + 1] = uniform_f32(&s, -100.0f, 100.0f);
        pos3[3 * i + 2] = 0.0f;
    }
}

/* ------------------------------------------------------------------------------------
 * Instance matrix (src/main.rs:437-439, rotation_of src/main.rs:141-143).
 * ---------------------------------------------------------------------------------- */
static void vec4_scale(float out[4], const float v[4], float s)
{
    for (int k = 0; k < 4; ++k) out[k] = v[k] * s;
}

static void instance_matrix(float m[16], const float p[3], const float v[3])
{
    float theta = atan2f(v[1], v[0]); /* main.rs:142: cgmath::Angle::atan2(v.y, v.x) */
    float s = sinf(theta);            /* f32::sin_cos = (sin, cos) */
    float c = cosf(theta);
    /* lhs = from_translation(p): columns a,b,c,d */
    const float a[4] = {1.0f, 0.0f, 0.0f, 0.0f};
    const float b[4] = {0.0f, 1.0f, 0.0f, 0.0f};
    const float cc[4] = {0.0f, 0.0f, 1.0f, 0.0f};
    const float d[4] = {p[0], p[1], p[2], 1.0f};
    /* rhs = from_angle_z(theta): columns */
    const float r[4][4] = {{c, s, 0.0f, 0.0f}, {-s, c, 0.0f, 0.0f}, {0.0f, 0.0f, 1.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 1.0f}};
    for (int k = 0; k < 4; ++k) {
        float t0[4], t1[4], t2[4], t3[4];
        vec4_scale(t0, a, r[k][0]);
        vec4_scale(t1, b, r[k][1]);
        vec4_scale(t2, cc, r[k][2]);
        vec4_scale(t3, d, r[k][3]);
        for (int e = 0; e < 4; ++e) m[4 * k + e] = ((t0[e] + t1[e]) + t2[e]) + t3[e];
    }
}

NBO_API void nbo_instances(const float *pos3, const float *vel3, float *inst16, uint32_t n)
{
    for (uint32_t i = 0; i < n; ++i) instance_matrix(inst16 + 16 * (size_t)i, pos3 + 3 * (size_t)i, vel3 + 3 * (size_t)i);
}

/* ------------------------------------------------------------------------------------
 * One step for bodies [first, first+count) against the start-of-step snapshot.
 *   old_pos3  : snapshot of ALL n_total positions   (main.rs:415 old_positions)
 *   pos3_out  : count new positions, local index (i - first)
 *   vel3      : count velocities, updated in place, local index
 *   inst16    : count matrices or NULL
 * The per-body closure is src/main.rs:424-440.
 * ---------------------------------------------------------------------------------- */
NBO_API void nbo_step_range(const float *old_pos3, float *pos3_out, float *vel3, float *inst16, uint32_t n_total,
                            uint32_t first, uint32_t count, float dt, float G, float bias)
{
    for (uint32_t l = 0; l < count; ++l) {
        const uint32_t n = first + l;
        /* boid_n_pos equals old_positions[n]: it was copied at main.rs:415 and is not yet written */
        const float pnx = old_pos3[3 * (size_t)n + 0];
        const float pny = old_pos3[3 * (size_t)n + 1];
        const float pnz = old_pos3[3 * (size_t)n + 2];
        /* main.rs:425-426: fold from Vector3::new(0.0, 0.0, 0.0), i = 0..n_total-1 in order, self included */
        float sx = 0.0f, sy = 0.0f, sz = 0.0f;
        for (uint32_t i = 0; i < n_total; ++i) {
            const float pix = old_pos3[3 * (size_t)i + 0];
            const float piy = old_pos3[3 * (size_t)i + 1];
            const float piz = old_pos3[3 * (size_t)i + 2];
            /* main.rs:428  vec = boid_i_pos.to_vec() - boid_n_pos.to_vec() */
            const float vx = pix - pnx, vy = piy - pny, vz = piz - pnz;
            /* main.rs:429  dist = boid_n_pos.distance2(*boid_i_pos) + bias; distance2 = (other - self).magnitude2() */
            const float ex = pix - pnx, ey = piy - pny, ez = piz - pnz;
            const float xx = ex * ex, yy = ey * ey, zz = ez * ez;
            const float d2 = (xx + yy) + zz;
            const float dist = d2 + bias;
            /* main.rs:430  sum + (vec * G / dist)  ==  sum + ((vec * G) / dist) */
            const float gx = vx * G, gy = vy * G, gz = vz * G;
            const float qx = gx / dist, qy = gy / dist, qz = gz / dist;
            sx = sx + qx;
            sy = sy + qy;
            sz = sz + qz;
        }
        /* main.rs:434  *boid_n_vel = *boid_n_vel + gravity * dt */
        float *v = vel3 + 3 * (size_t)l;
        const float ax = sx * dt, ay = sy * dt, az = sz * dt;
        v[0] = v[0] + ax;
        v[1] = v[1] + ay;
        v[2] = v[2] + az;
        /* main.rs:436  *boid_n_pos = Point3::from_vec(boid_n_vel.clone() + boid_n_pos.to_vec())   (no dt) */
        float *p = pos3_out + 3 * (size_t)l;
        p[0] = v[0] + pnx;
        p[1] = v[1] + pny;
        p[2] = v[2] + pnz;
        /* main.rs:437-439 */
        if (inst16) instance_matrix(inst16 + 16 * (size_t)l, p, v);
    }
}

/* ------------------------------------------------------------------------------------
 * The same step, eight bodies at a time (x86 AVX2; anything else takes the loop above).
 * Each vector lane is one body n and performs exactly the scalar operations of nbo_step_range on
 * its own operands, in the same order over i (IEEE add / sub / mul / div per lane; no contraction,
 * no reassociation), so the results are bit-identical by construction; tests/test_oracle.py holds
 * the two to equality.  It exists because the scalar loop needs hours for the headline size
 * (N = 131 072, 1 000 steps: tests/golden/make_golden.py --c3); the CPU baseline of bench.py
 * times the SCALAR loop, which is what the reference's rayon closure compiles to.
 * ---------------------------------------------------------------------------------- */
#if defined(__x86_64__)
typedef float nbo_v8 __attribute__((vector_size(32)));

__attribute__((target("avx2"))) static void step_range_v8(const float *old_pos3, float *pos3_out, float *vel3,
                                                          uint32_t n_total, uint32_t first, uint32_t count, float dt,
                                                          float G, float bias)
{
    const nbo_v8 vG = {G, G, G, G, G, G, G, G}, vbias = {bias, bias, bias, bias, bias, bias, bias, bias};
    const nbo_v8 vdt = {dt, dt, dt, dt, dt, dt, dt, dt};
    for (uint32_t l0 = 0; l0 + 8 <= count; l0 += 8) {
        nbo_v8 pnx, pny, pnz, sx = {0}, sy = {0}, sz = {0};
        for (int k = 0; k < 8; ++k) {
            const size_t n = (size_t)first + l0 + (uint32_t)k;
            pnx[k] = old_pos3[3 * n + 0];
            pny[k] = old_pos3[3 * n + 1];
            pnz[k] = old_pos3[3 * n + 2];
        }
        for (uint32_t i = 0; i < n_total; ++i) {
            const float fx = old_pos3[3 * (size_t)i + 0], fy = old_pos3[3 * (size_t)i + 1], fz = old_pos3[3 * (size_t)i + 2];
            const nbo_v8 pix = {fx, fx, fx, fx, fx, fx, fx, fx}, piy = {fy, fy, fy, fy, fy, fy, fy, fy},
                         piz = {fz, fz, fz, fz, fz, fz, fz, fz};
            const nbo_v8 vx = pix - pnx, vy = piy - pny, vz = piz - pnz; /* main.rs:428 (and the difference inside distance2) */
            const nbo_v8 xx = vx * vx, yy = vy * vy, zz = vz * vz;
            const nbo_v8 d2 = (xx + yy) + zz;
            const nbo_v8 dist = d2 + vbias; /* main.rs:429 */
            const nbo_v8 gx = vx * vG, gy = vy * vG, gz = vz * vG;
            sx = sx + gx / dist; /* main.rs:430 */
            sy = sy + gy / dist;
            sz = sz + gz / dist;
        }
        const nbo_v8 ax = sx * vdt, ay = sy * vdt, az = sz * vdt;
        for (int k = 0; k < 8; ++k) {
            float *v = vel3 + 3 * (size_t)(l0 + (uint32_t)k);
            float *p = pos3_out + 3 * (size_t)(l0 + (uint32_t)k);
            v[0] = v[0] + ax[k]; /* main.rs:434 */
            v[1] = v[1] + ay[k];
            v[2] = v[2] + az[k];
            p[0] = v[0] + pnx[k]; /* main.rs:436 */
            p[1] = v[1] + pny[k];
            p[2] = v[2] + pnz[k];
        }
    }
}
#endif

/* 1 when nbo_step_range_batched really runs eight bodies per vector on this CPU */
NBO_API int nbo_batched_available(void)
{
#if defined(__x86_64__)
    return __builtin_cpu_supports("avx2") ? 1 : 0;
#else
    return 0;
#endif
}

NBO_API void nbo_step_range_batched(const float *old_pos3, float *pos3_out, float *vel3, float *inst16, uint32_t n_total,
                                    uint32_t first, uint32_t count, float dt, float G, float bias)
{
    uint32_t done = 0;
#if defined(__x86_64__)
    if (__builtin_cpu_supports("avx2")) {
        done = count & ~7u;
        step_range_v8(old_pos3, pos3_out, vel3, n_total, first, done, dt, G, bias);
        if (inst16)
            for (uint32_t l = 0; l < done; ++l)
                instance_matrix(inst16 + 16 * (size_t)l, pos3_out + 3 * (size_t)l, vel3 + 3 * (size_t)l);
    }
#endif
    nbo_step_range(old_pos3, pos3_out + 3 * (size_t)done, vel3 + 3 * (size_t)done, inst16 ? inst16 + 16 * (size_t)done : NULL,
                   n_total, first + done, count - done, dt, G, bias);
}

/* The outer loop over bodies is rayon's par_iter_mut (main.rs:420-424).  Bodies are
 * independent within a step, so the thread count changes scheduling only, never bits.
 * Threads live for the whole run and meet at a barrier twice per step (after the snapshot
 * copy, after the update). */
typedef struct {
    float *old_pos3;
    float *pos3;
    float *vel3;
    float *inst16;
    uint32_t n, k, first, count;
    float dt, G, bias;
    int tid;
    pthread_barrier_t *bar;
    int batched;
} nbo_job;

static void *nbo_worker(void *arg)
{
    nbo_job *j = (nbo_job *)arg;
    for (uint32_t s = 0; s < j->k; ++s) {
        /* main.rs:415  old_positions.copy_from_slice(positions); main.rs:416 also copies the
         * velocities, but update_instance_nbody never reads that copy */
        if (j->tid == 0) memcpy(j->old_pos3, j->pos3, sizeof(float) * 3 * (size_t)j->n);
        if (j->bar) pthread_barrier_wait(j->bar);
        float *inst = (j->inst16 && s + 1 == j->k) ? j->inst16 + 16 * (size_t)j->first : NULL;
        (j->batched ? nbo_step_range_batched : nbo_step_range)(j->old_pos3, j->pos3 + 3 * (size_t)j->first,
                                                                j->vel3 + 3 * (size_t)j->first, inst, j->n, j->first,
                                                                j->count, j->dt, j->G, j->bias);
        if (j->bar) pthread_barrier_wait(j->bar);
    }
    return NULL;
}

/* update_instance_nbody (main.rs:404-441) applied k times.  inst16 may be NULL; when given it
 * holds the matrices of the LAST step, as the caller at main.rs:932-936 would see them. */
static int run_impl(float *pos3, float *vel3, float *inst16, uint32_t n, uint32_t k, float dt, float G, float bias,
                    int nthreads, int batched)
{
    float *old = (float *)malloc(sizeof(float) * 3 * (size_t)(n ? n : 1));
    if (!old) return -1;
    if (nthreads < 1) nthreads = 1;
    if ((uint32_t)nthreads > n) nthreads = n ? (int)n : 1;
    if (nthreads == 1) {
        nbo_job job = {old, pos3, vel3, inst16, n, k, 0, n, dt, G, bias, 0, NULL, batched};
        nbo_worker(&job);
        free(old);
        return 0;
    }
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, (unsigned)nthreads);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    nbo_job *jobs = (nbo_job *)malloc(sizeof(nbo_job) * (size_t)nthreads);
    for (int t = 0; t < nthreads; ++t) {
        uint32_t lo = (uint32_t)(((uint64_t)n * (uint64_t)t) / (uint64_t)nthreads);
        uint32_t hi = (uint32_t)(((uint64_t)n * (uint64_t)(t + 1)) / (uint64_t)nthreads);
        jobs[t] = (nbo_job){old, pos3, vel3, inst16, n, k, lo, hi - lo, dt, G, bias, t, &bar, batched};
        pthread_create(&th[t], NULL, nbo_worker, &jobs[t]);
    }
    for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    pthread_barrier_destroy(&bar);
    free(jobs);
    free(th);
    free(old);
    return 0;
}

NBO_API int nbo_run(float *pos3, float *vel3, float *inst16, uint32_t n, uint32_t k, float dt, float G, float bias,
                    int nthreads)
{
    return run_impl(pos3, vel3, inst16, n, k, dt, G, bias, nthreads, 0);
}

/* nbo_run through nbo_step_range_batched (same bits; see there) */
NBO_API int nbo_run_batched(float *pos3, float *vel3, float *inst16, uint32_t n, uint32_t k, float dt, float G, float bias,
                            int nthreads)
{
    return run_impl(pos3, vel3, inst16, n, k, dt, G, bias, nthreads, 1);
}

/* ====================================================================================
 * Boids controller: update_instance_boids, src/main.rs:443-526 --
 * the controller the reference's event loop actually calls (src/main.rs:925).
 *
 * cgmath semantics used here, besides those in the header:
 *   MetricSpace::distance(self, other)   sqrt(distance2(self, other))          (f32::sqrt: correctly rounded)
 *   Vector3::magnitude()                 sqrt(dot(self, self))
 *   Vector3::normalize_to(m)             self * (m / self.magnitude())
 * Integer literals `0` / `cnt + 1` are i32; `count as f32` is an exact conversion for counts < 2^24.
 * ==================================================================================== */
typedef struct nbo_boids_params {
    float dt;               /* 0.04   main.rs:450 */
    float rule_1_distance;  /* 1000.0 main.rs:451  (compared with the SQUARED distance, main.rs:474-475) */
    float rule_2_distance;  /* 5.0    main.rs:452 */
    float rule_3_distance;  /* 500.0  main.rs:453 */
    float rule_1_scale;     /* 0.02   main.rs:454 */
    float rule_2_scale;     /* 0.05   main.rs:455 */
    float rule_3_scale;     /* 0.5    main.rs:456 */
} nbo_boids_params;

NBO_API void nbo_boids_default_params(nbo_boids_params *p)
{
    p->dt = 0.04f;
    p->rule_1_distance = 1000.0f;
    p->rule_2_distance = 5.0f;
    p->rule_3_distance = 500.0f;
    p->rule_1_scale = 0.02f;
    p->rule_2_scale = 0.05f;
    p->rule_3_scale = 0.5f;
}

static float dist2_f32(const float *a, const float *b) /* a.distance2(b) = (b - a).magnitude2() */
{
    const float ex = b[0] - a[0], ey = b[1] - a[1], ez = b[2] - a[2];
    const float xx = ex * ex, yy = ey * ey, zz = ez * ez;
    return (xx + yy) + zz;
}

/* One boids step for bodies [first, first+count) against the snapshots of ALL positions and velocities
 * (old_positions / old_velocities, main.rs:459-460).  pos3_out, vel3 (in: this range's velocities, out: new) and
 * inst16 use the local index (n - first).
 * The two snapshots have their own lengths: the position folds run over old_positions.iter() (main.rs:471, 482), the
 * velocity fold over old_velocities.iter() (main.rs:494) -- nothing in the reference indexes one by the other's length,
 * and the zip (main.rs:465-469) only bounds which bodies n are updated (first + count <= min of the lengths). */
NBO_API void nbo_boids_step_range2(const float *old_pos3, uint32_t n_total, const float *old_vel3, uint32_t n_total_vel,
                                   float *pos3_out, float *vel3, float *inst16, uint32_t first, uint32_t count,
                                   const nbo_boids_params *bp)
{
    for (uint32_t l = 0; l < count; ++l) {
        const uint32_t n = first + l;
        const float *pn = old_pos3 + 3 * (size_t)n; /* boid_n_pos == old_positions[n] until written at main.rs:521 */
        float *v = vel3 + 3 * (size_t)l;            /* boid_n_vel == old_velocities[n] until written at main.rs:514 */
        /* main.rs:471-480  (flock_center, count) */
        float cx = 0.0f, cy = 0.0f, cz = 0.0f;
        int32_t cnt = 0;
        for (uint32_t i = 0; i < n_total; ++i) {
            const float *pi = old_pos3 + 3 * (size_t)i;
            const float dist = dist2_f32(pn, pi);                  /* main.rs:474 distance2: squared */
            if (dist < bp->rule_1_distance && n != i) {            /* main.rs:475 */
                cx = cx + pi[0];                                   /* main.rs:476 sum + boid_i_pos.to_vec() */
                cy = cy + pi[1];
                cz = cz + pi[2];
                cnt = cnt + 1;
            }
        }
        /* main.rs:482-492  flock_repel */
        float rx = 0.0f, ry = 0.0f, rz = 0.0f;
        for (uint32_t i = 0; i < n_total; ++i) {
            const float *pi = old_pos3 + 3 * (size_t)i;
            const float dist = sqrtf(dist2_f32(pn, pi));           /* main.rs:485 distance */
            if (dist < bp->rule_2_distance && n != i) {            /* main.rs:486 */
                const float wx = pi[0] - pn[0], wy = pi[1] - pn[1], wz = pi[2] - pn[2];
                rx = rx - wx;                                      /* main.rs:487 sum - (p_i - p_n) */
                ry = ry - wy;
                rz = rz - wz;
            }
        }
        /* main.rs:494-504  (flock_match, vcount) over the velocity snapshot */
        float mx = 0.0f, my = 0.0f, mz = 0.0f;
        int32_t vcnt = 0;
        for (uint32_t i = 0; i < n_total_vel; ++i) {
            const float *vi = old_vel3 + 3 * (size_t)i;
            const float dist = sqrtf(dist2_f32(v, vi));            /* main.rs:497 boid_n_vel.distance(*boid_i_vel) */
            if (dist < bp->rule_3_distance && n != i) {            /* main.rs:498 */
                mx = mx + vi[0];                                   /* main.rs:499 */
                my = my + vi[1];
                mz = mz + vi[2];
                vcnt = vcnt + 1;
            }
        }
        if (cnt > 0) { /* main.rs:506-508 */
            const float c = (float)cnt;
            cx = cx / c;
            cy = cy / c;
            cz = cz / c;
        }
        if (vcnt > 0) { /* main.rs:510-512 */
            const float c = (float)vcnt;
            mx = mx / c;
            my = my / c;
            mz = mz / c;
        }
        /* main.rs:514  vel = center*s1 + repel*s2 + match*s3   (left to right) */
        float nvx = (cx * bp->rule_1_scale + rx * bp->rule_2_scale) + mx * bp->rule_3_scale;
        float nvy = (cy * bp->rule_1_scale + ry * bp->rule_2_scale) + my * bp->rule_3_scale;
        float nvz = (cz * bp->rule_1_scale + rz * bp->rule_2_scale) + mz * bp->rule_3_scale;
        /* main.rs:516-518  speed clamp */
        const float mag = sqrtf((nvx * nvx + nvy * nvy) + nvz * nvz);
        if (mag > 1.0f) {
            const float s = 1.0f / mag; /* normalize_to(1.0) = self * (1.0 / magnitude) */
            nvx = nvx * s;
            nvy = nvy * s;
            nvz = nvz * s;
        }
        v[0] = nvx;
        v[1] = nvy;
        v[2] = nvz;
        /* main.rs:521  pos = vel * dt + pos */
        float *p = pos3_out + 3 * (size_t)l;
        const float sx = nvx * bp->dt, sy = nvy * bp->dt, sz = nvz * bp->dt;
        p[0] = sx + pn[0];
        p[1] = sy + pn[1];
        p[2] = sz + pn[2];
        /* main.rs:522-524 */
        if (inst16) instance_matrix(inst16 + 16 * (size_t)l, p, v);
    }
}

NBO_API void nbo_boids_step_range(const float *old_pos3, const float *old_vel3, float *pos3_out, float *vel3, float *inst16,
                                  uint32_t n_total, uint32_t first, uint32_t count, const nbo_boids_params *bp)
{
    nbo_boids_step_range2(old_pos3, n_total, old_vel3, n_total, pos3_out, vel3, inst16, first, count, bp);
}

typedef struct {
    float *old_pos3, *old_vel3, *pos3, *vel3, *inst16;
    uint32_t n, k, first, count;
    const nbo_boids_params *bp;
    int tid;
    pthread_barrier_t *bar;
} nbo_boids_job;

static void *nbo_boids_worker(void *arg)
{
    nbo_boids_job *j = (nbo_boids_job *)arg;
    for (uint32_t s = 0; s < j->k; ++s) {
        if (j->tid == 0) { /* main.rs:459-460 */
            memcpy(j->old_pos3, j->pos3, sizeof(float) * 3 * (size_t)j->n);
            memcpy(j->old_vel3, j->vel3, sizeof(float) * 3 * (size_t)j->n);
        }
        if (j->bar) pthread_barrier_wait(j->bar);
        float *inst = (j->inst16 && s + 1 == j->k) ? j->inst16 + 16 * (size_t)j->first : NULL;
        nbo_boids_step_range(j->old_pos3, j->old_vel3, j->pos3 + 3 * (size_t)j->first, j->vel3 + 3 * (size_t)j->first, inst,
                             j->n, j->first, j->count, j->bp);
        if (j->bar) pthread_barrier_wait(j->bar);
    }
    return NULL;
}

/* update_instance_boids (main.rs:443-526) applied k times. */
NBO_API int nbo_boids_run(float *pos3, float *vel3, float *inst16, uint32_t n, uint32_t k, const nbo_boids_params *bp,
                          int nthreads)
{
    float *oldp = (float *)malloc(sizeof(float) * 3 * (size_t)(n ? n : 1));
    float *oldv = (float *)malloc(sizeof(float) * 3 * (size_t)(n ? n : 1));
    if (!oldp || !oldv) return -1;
    if (nthreads < 1) nthreads = 1;
    if ((uint32_t)nthreads > n) nthreads = n ? (int)n : 1;
    if (nthreads == 1) {
        nbo_boids_job job = {oldp, oldv, pos3, vel3, inst16, n, k, 0, n, bp, 0, NULL};
        nbo_boids_worker(&job);
    } else {
        pthread_barrier_t bar;
        pthread_barrier_init(&bar, NULL, (unsigned)nthreads);
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
        nbo_boids_job *jobs = (nbo_boids_job *)malloc(sizeof(nbo_boids_job) * (size_t)nthreads);
        for (int t = 0; t < nthreads; ++t) {
            uint32_t lo = (uint32_t)(((uint64_t)n * (uint64_t)t) / (uint64_t)nthreads);
            uint32_t hi = (uint32_t)(((uint64_t)n * (uint64_t)(t + 1)) / (uint64_t)nthreads);
            jobs[t] = (nbo_boids_job){oldp, oldv, pos3, vel3, inst16, n, k, lo, hi - lo, bp, t, &bar};
            pthread_create(&th[t], NULL, nbo_boids_worker, &jobs[t]);
        }
        for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
        pthread_barrier_destroy(&bar);
        free(jobs);
        free(th);
    }
    free(oldp);
    free(oldv);
    return 0;
}

/* ====================================================================================
 * CameraArray::update, src/gfx.rs:397-408 with build_camera src/gfx.rs:358-369:
 * per entity  M = (OPENGL_TO_WGPU_MATRIX * perspective(fov, aspect, 1, 10000)) * look_at_dir(eye, dir, up).
 * `correction * proj * view` associates to the left, so correction * proj is one constant matrix for the whole
 * array: the caller passes it (16 floats, column-major) -- its tan() stays with the caller's cgmath::perspective.
 * cgmath 0.17 semantics restated:
 *   look_at_dir(eye, dir, up): f = dir.normalize(); s = f.cross(up).normalize(); u = s.cross(f);
 *       columns (s.x, u.x, -f.x, 0), (s.y, u.y, -f.y, 0), (s.z, u.z, -f.z, 0), (-eye.dot(s), -eye.dot(u), eye.dot(f), 1)
 *   normalize() = self * (1 / magnitude());  cross = (y*oz - z*oy, z*ox - x*oz, x*oy - y*ox);  dot = (xx + yy) + zz
 *   Matrix4 * Matrix4 as in instance_matrix above.
 * ==================================================================================== */
static void normalize3(float v[3])
{
    const float mag = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    const float s = 1.0f / mag;
    v[0] = v[0] * s;
    v[1] = v[1] * s;
    v[2] = v[2] * s;
}

static void cross3(float out[3], const float a[3], const float b[3])
{
    out[0] = a[1] * b[2] - a[2] * b[1];
    out[1] = a[2] * b[0] - a[0] * b[2];
    out[2] = a[0] * b[1] - a[1] * b[0];
}

static float dot3(const float a[3], const float b[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

NBO_API void nbo_cameras(const float *eye3, const float *dir3, const float *up3, const float *cp16, float *out16, uint32_t n)
{
    for (uint32_t i = 0; i < n; ++i) {
        const float *eye = eye3 + 3 * (size_t)i;
        float f[3] = {dir3[3 * (size_t)i], dir3[3 * (size_t)i + 1], dir3[3 * (size_t)i + 2]};
        normalize3(f);
        float s[3], u[3];
        cross3(s, f, up3);
        normalize3(s);
        cross3(u, s, f);
        const float view[4][4] = {{s[0], u[0], -f[0], 0.0f},
                                  {s[1], u[1], -f[1], 0.0f},
                                  {s[2], u[2], -f[2], 0.0f},
                                  {-dot3(eye, s), -dot3(eye, u), dot3(eye, f), 1.0f}};
        float *m = out16 + 16 * (size_t)i;
        for (int k = 0; k < 4; ++k)      /* column k of cp * view */
            for (int e = 0; e < 4; ++e) {
                const float t0 = cp16[0 + e] * view[k][0], t1 = cp16[4 + e] * view[k][1];
                const float t2 = cp16[8 + e] * view[k][2], t3 = cp16[12 + e] * view[k][3];
                m[4 * k + e] = ((t0 + t1) + t2) + t3;
            }
    }
}

/* The array's constant  OPENGL_TO_WGPU_MATRIX * perspective(vertical_fov, aspect, near, far)  of build_camera
 * (src/gfx.rs:365, 367; OPENGL_TO_WGPU_MATRIX src/gfx.rs:12-17; the reference passes near = 1, far = 10000 and, from
 * CameraArray::new src/gfx.rs:381, vertical_fov = horizontal_fov / aspect_ratio).  cgmath 0.17 restated:
 *   Deg -> Rad: deg * (PI / 180) with the constant formed in binary64 and cast;  perspective(fovy, aspect, near, far) =
 *   PerspectiveFov -> Matrix4: f = cot(fovy / 2) = 1 / tan(fovy / 2);  c0r0 = f / aspect, c1r1 = f,
 *   c2r2 = (far + near) / (near - far), c2r3 = -1, c3r2 = (2 * far * near) / (near - far), every other entry 0.
 * Returns 0, or -1 where cgmath's assertions would panic (fovy outside (0, 180) degrees, aspect == 0, near <= 0, far <= near...). */
NBO_API int nbo_camera_constant(float vertical_fov_deg, float aspect, float near, float far, float *cp16)
{
    const float fovy = vertical_fov_deg * (float)(3.14159265358979323846 / 180.0);
    if (!(fovy > 0.0f) || !(fovy < (float)3.14159265358979323846) || !(aspect != 0.0f) || !(near > 0.0f) || !(far > 0.0f) ||
        !(far != near))
        return -1;
    const float f = 1.0f / tanf(fovy / 2.0f);
    float proj[4][4] = {{0}};  /* proj[k] = column k */
    proj[0][0] = f / aspect;
    proj[1][1] = f;
    proj[2][2] = (far + near) / (near - far);
    proj[2][3] = -1.0f;
    proj[3][2] = ((2.0f * far) * near) / (near - far);
    static const float corr[4][4] = {{1.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 1.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.5f, 0.0f}, {0.0f, 0.0f, 0.5f, 1.0f}};
    for (int k = 0; k < 4; ++k)
        for (int e = 0; e < 4; ++e) {
            const float t0 = corr[0][e] * proj[k][0], t1 = corr[1][e] * proj[k][1];
            const float t2 = corr[2][e] * proj[k][2], t3 = corr[3][e] * proj[k][3];
            cp16[4 * k + e] = ((t0 + t1) + t2) + t3;
        }
    return 0;
}

/* ====================================================================================
 * Random-walk controller: update_instance_random, src/main.rs:381-402.
 *   vel += (U[-0.0001, 0.0001), U[-0.0001, 0.0001), 0);  pos += vel;  matrix as above.
 * The reference draws from the unseeded thread_rng of whichever rayon worker runs the body, so only the
 * distribution can be kept; the stream here is build-owned and counter based: the two draws of body n at step t come
 * from splitmix64 seeded with (seed, t, n), so any sharding and any execution order give the same bits.
 * ==================================================================================== */
static float random_jitter(uint64_t seed, uint64_t step, uint32_t body, uint32_t which)
{
    uint64_t s = seed ^ (step * 0xD1B54A32D192ED03ull) ^ ((uint64_t)body * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)which << 62);
    return uniform_f32(&s, -0.0001f, 0.0001f);
}

NBO_API void nbo_random_step_range(float *pos3, float *vel3, float *inst16, uint32_t first, uint32_t count, uint64_t seed,
                                   uint64_t step)
{
    for (uint32_t l = 0; l < count; ++l) {
        float *v = vel3 + 3 * (size_t)l, *p = pos3 + 3 * (size_t)l;
        const float jx = random_jitter(seed, step, first + l, 0), jy = random_jitter(seed, step, first + l, 1);
        v[0] = v[0] + jx; /* main.rs:392-396  *vel += Vector3::new(jx, jy, 0.0) */
        v[1] = v[1] + jy;
        v[2] = v[2] + 0.0f;
        p[0] = p[0] + v[0]; /* main.rs:397  *pos += *vel */
        p[1] = p[1] + v[1];
        p[2] = p[2] + v[2];
        if (inst16) instance_matrix(inst16 + 16 * (size_t)l, p, v); /* main.rs:398-400 */
    }
}

/* ------------------------------------------------------------------------------------
 * binary64 variant of the same recurrence: NOT the reference's arithmetic -- it exists to
 * show the rounding-noise floor of the chaotic system (sixth finding).
 * ---------------------------------------------------------------------------------- */
NBO_API int nbo_run_f64(double *pos3, double *vel3, uint32_t n, uint32_t k, double dt, double G, double bias)
{
    double *old = (double *)malloc(sizeof(double) * 3 * (size_t)(n ? n : 1));
    if (!old) return -1;
    for (uint32_t s = 0; s < k; ++s) {
        memcpy(old, pos3, sizeof(double) * 3 * (size_t)n);
        for (uint32_t b = 0; b < n; ++b) {
            double sx = 0, sy = 0, sz = 0;
            const double px = old[3 * (size_t)b], py = old[3 * (size_t)b + 1], pz = old[3 * (size_t)b + 2];
            for (uint32_t i = 0; i < n; ++i) {
                const double vx = old[3 * (size_t)i] - px, vy = old[3 * (size_t)i + 1] - py, vz = old[3 * (size_t)i + 2] - pz;
                const double dist = ((vx * vx + vy * vy) + vz * vz) + bias;
                sx = sx + (vx * G) / dist;
                sy = sy + (vy * G) / dist;
                sz = sz + (vz * G) / dist;
            }
            double *v = vel3 + 3 * (size_t)b, *p = pos3 + 3 * (size_t)b;
            v[0] = v[0] + sx * dt;
            v[1] = v[1] + sy * dt;
            v[2] = v[2] + sz * dt;
            p[0] = v[0] + px;
            p[1] = v[1] + py;
            p[2] = v[2] + pz;
        }
    }
    free(old);
    return 0;
}

/* The same binary64 recurrence for bodies [first, first+count) of ONE step against a binary32 snapshot (each record
 * widened exactly): the velocity change a_n*dt with ~1e-16 relative error per term, i.e. the yardstick against which the
 * rounding error of the reference's own binary32 sum (and of any reassociated binary32 sum) is measured in the tests.
 * dv3_out: 3*count doubles = the step's velocity change of each body. */
NBO_API void nbo_step_range_dv_f64(const float *old_pos3, double *dv3_out, uint32_t n_total, uint32_t first, uint32_t count,
                                   double dt, double G, double bias)
{
    for (uint32_t l = 0; l < count; ++l) {
        const size_t b = (size_t)first + l;
        const double px = old_pos3[3 * b], py = old_pos3[3 * b + 1], pz = old_pos3[3 * b + 2];
        double sx = 0, sy = 0, sz = 0;
        for (uint32_t i = 0; i < n_total; ++i) {
            const double vx = (double)old_pos3[3 * (size_t)i] - px, vy = (double)old_pos3[3 * (size_t)i + 1] - py,
                         vz = (double)old_pos3[3 * (size_t)i + 2] - pz;
            const double dist = ((vx * vx + vy * vy) + vz * vz) + bias;
            sx = sx + (vx * G) / dist;
            sy = sy + (vy * G) / dist;
            sz = sz + (vz * G) / dist;
        }
        dv3_out[3 * (size_t)l] = sx * dt;
        dv3_out[3 * (size_t)l + 1] = sy * dt;
        dv3_out[3 * (size_t)l + 2] = sz * dt;
    }
}
