// ubench_mfma_mix.hip -- does the f32 matrix pipe run BESIDE the vector ALU on gfx950, and is D = A*1 + C an exact subtract?
// (diagnostic tool, not product code)
//
// Questions it answers, for the pair-fold kernels' design:
//   1. throughput of a stream of F v_fma_f32 + M v_mfma_f32_{4x4x1_16b,16x16x4,32x32x2}_f32 per loop trip as a function of
//      waves per SIMD: if the trip costs max(vector, matrix) the two pipes overlap, if it costs the sum they do not, and the
//      slope in M gives what one MFMA costs the VECTOR issue port;
//   2. is v_mfma_f32_4x4x1_16b_f32 with A = x_j (broadcast from one block: cbsz = 4, abid = b), B = 1.0, C = -x_i bit-identical
//      to v_sub_f32 x_j - x_i for every lane and every register, over wide-exponent random data, signed zeros, subnormals,
//      infinities and NaNs.
//
// Build: hipcc -O2 --offload-arch=gfx950 -o build/ubench_mfma_mix tools/ubench_mfma_mix.hip
// Run:   build/ubench_mfma_mix
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

enum Shape { S_NONE = 0, S_4x4x1 = 1, S_16x16x4 = 2, S_32x32x2 = 3 };

// F independent v_fma_f32 chains + M MFMAs (independent accumulators) per trip
template <int F, int M, int SHAPE>
__global__ __launch_bounds__(256) void mix(uint64_t *out, float *sinkp, int iters, float seed)
{
    const float l = (float)(threadIdx.x & 63) * 1e-3f;
    float a[F > 0 ? F : 1];
#pragma unroll
    for (int k = 0; k < (F > 0 ? F : 1); ++k) a[k] = seed + l + (float)k;
    const float b = 1.0000001f, c = 1e-9f;
    f4 acc4[M > 0 ? M : 1];
    f16v acc16[(SHAPE == S_32x32x2 && M > 0) ? M : 1];
#pragma unroll
    for (int k = 0; k < (M > 0 ? M : 1); ++k) acc4[k] = f4{l, l, l, l};
#pragma unroll
    for (int k = 0; k < ((SHAPE == S_32x32x2 && M > 0) ? M : 1); ++k)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc16[k][e] = l;
    float ma = seed * 0.5f + l, mb = 1.0f;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            // interleave: each MFMA is followed by its share of the FMAs
            constexpr int kPer = M > 0 ? (F + M - 1) / M : F;
#pragma unroll
            for (int m = 0; m < (M > 0 ? M : 1); ++m) {
                if constexpr (M > 0) {
                    if constexpr (SHAPE == S_4x4x1) acc4[m] = __builtin_amdgcn_mfma_f32_4x4x1f32(ma, mb, acc4[m], 4, 3, 0);
                    if constexpr (SHAPE == S_16x16x4) acc4[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ma, mb, acc4[m], 0, 0, 0);
                    if constexpr (SHAPE == S_32x32x2) acc16[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(ma, mb, acc16[m], 0, 0, 0);
                }
#pragma unroll
                for (int k = m * kPer; k < (m + 1) * kPer && k < F; ++k) {
                    a[k] = __builtin_fmaf(a[k], b, c);
                    asm volatile("" : "+v"(a[k]));
                }
            }
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint64_t r1 = __builtin_amdgcn_s_memrealtime();
    float sink = 0.f;
#pragma unroll
    for (int k = 0; k < (F > 0 ? F : 1); ++k) sink += a[k];
#pragma unroll
    for (int k = 0; k < (M > 0 ? M : 1); ++k) sink += acc4[k][0] + acc4[k][1] + acc4[k][2] + acc4[k][3];
#pragma unroll
    for (int k = 0; k < ((SHAPE == S_32x32x2 && M > 0) ? M : 1); ++k)
#pragma unroll
        for (int e = 0; e < 16; ++e) sink += acc16[k][e];
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) {
        out[2 * wave] = t1 - t0;
        out[2 * wave + 1] = r1 - r0;
    }
    if (sink == 123.456f) sinkp[0] = sink;
}

template <int F, int M, int SHAPE>
static void run(int waves_per_simd, int iters)
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * waves_per_simd;
    const int waves = blocks * 4;
    uint64_t *d;
    float *sinkp;
    hipMalloc(&d, sizeof(uint64_t) * 2 * waves);
    hipMalloc(&sinkp, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((mix<F, M, SHAPE>), dim3(blocks), dim3(256), 0, 0, d, sinkp, iters / 10 + 1, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((mix<F, M, SHAPE>), dim3(blocks), dim3(256), 0, 0, d, sinkp, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(2 * waves);
    hipMemcpy(h.data(), d, sizeof(uint64_t) * 2 * waves, hipMemcpyDeviceToHost);
    std::vector<double> cyc(waves), clk(waves);
    for (int w = 0; w < waves; ++w) {
        cyc[w] = (double)h[2 * w];
        clk[w] = (double)h[2 * w] / (double)h[2 * w + 1] * 100.0;
    }
    std::sort(cyc.begin(), cyc.end());
    std::sort(clk.begin(), clk.end());
    const double trips = (double)iters * 4.0;
    const double cyc_per_trip_wave = cyc[waves / 2] / trips;
    const double cyc_per_trip_simd = cyc_per_trip_wave / waves_per_simd;
    const double wall_cyc_per_trip_simd = (ms * 1e-3) * (clk[waves / 2] * 1e6) / trips / waves_per_simd;
    static const char *sn[] = {"none", "4x4x1_16b", "16x16x4", "32x32x2"};
    static const int pipe_cyc[] = {0, 8, 32, 64};
    printf("F=%2d fma + M=%2d mfma_%-10s waves/SIMD=%d  cyc/trip(SIMD)=%7.2f  wall-cyc/trip(SIMD)=%7.2f  [vector alone %3d, matrix alone %4d]  "
           "clock=%5.0f MHz  wall=%.3f ms\n",
           F, M, sn[SHAPE], waves_per_simd, cyc_per_trip_simd, wall_cyc_per_trip_simd, 2 * F, pipe_cyc[SHAPE] * M, clk[waves / 2], ms);
    fflush(stdout);
    hipFree(d);
    hipFree(sinkp);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

// ---- exactness: D[r] = x_{4b'+r} * 1 + (-xi) against x_{4b'+r} - xi, cbsz = 4 (A broadcast from block abid to all 16 blocks) ----
template <int ABID>
__device__ __forceinline__ f4 sub4(float xs, float negxi)
{
    return __builtin_amdgcn_mfma_f32_4x4x1f32(xs, 1.0f, f4{negxi, negxi, negxi, negxi}, 4, ABID, 0);
}

__global__ __launch_bounds__(64) void exact_kernel(const float *__restrict__ xj, const float *__restrict__ xi, uint32_t groups,
                                                   unsigned long long *bad, float *first_bad)
{
    const int lane = threadIdx.x;
    for (uint32_t g = blockIdx.x; g < groups; g += gridDim.x) {
        const float xs = xj[(size_t)g * 64 + lane];  // source j0 + lane
        const float me = xi[(size_t)g * 64 + lane];  // this lane's body
        const float neg = -me;
        f4 d[16];
#define SUB(b) d[b] = sub4<b>(xs, neg);
        SUB(0) SUB(1) SUB(2) SUB(3) SUB(4) SUB(5) SUB(6) SUB(7) SUB(8) SUB(9) SUB(10) SUB(11) SUB(12) SUB(13) SUB(14) SUB(15)
#undef SUB
#pragma unroll
        for (int b = 0; b < 16; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float src = __shfl(xs, 4 * b + r, 64);
                const float want = src - me;
                const float got = d[b][r];
                const uint32_t wu = __float_as_uint(want), gu = __float_as_uint(got);
                const bool both_nan = (want != want) && (got != got);
                if (wu != gu && !both_nan) {
                    if (atomicAdd(bad, 1ull) == 0ull) {
                        first_bad[0] = src;
                        first_bad[1] = me;
                        first_bad[2] = want;
                        first_bad[3] = got;
                    }
                }
            }
    }
}

static uint64_t sm64(uint64_t &s)
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static void exactness(uint32_t groups, int flavour)
{
    std::vector<float> a((size_t)groups * 64), b((size_t)groups * 64);
    uint64_t s = 1234 + flavour;
    const float specials[] = {0.f, -0.f, 1e-45f, -1e-45f, 1.1754942e-38f, 1.17549435e-38f, 3.4028235e38f, -3.4028235e38f,
                              __builtin_inff(), -__builtin_inff(), __builtin_nanf(""), 1.f, -1.f, 100.f, 99.99999f, 100.00001f};
    for (size_t i = 0; i < a.size(); ++i) {
        auto draw = [&]() -> float {
            uint64_t r = sm64(s);
            if (flavour == 0) {  // the bench's own range: U[-100, 100)
                return -100.f + 200.f * (float)(r >> 40) * 0x1.0p-24f;
            } else if (flavour == 1) {  // any bit pattern
                uint32_t u = (uint32_t)r;
                float f;
                memcpy(&f, &u, 4);
                return f;
            } else if (flavour == 2) {  // close values: cancellation
                float base = -100.f + 200.f * (float)((r >> 40) & 0xffff) * 0x1.0p-16f;
                uint32_t u;
                memcpy(&u, &base, 4);
                u += (uint32_t)((r >> 8) & 0x3f);
                float f;
                memcpy(&f, &u, 4);
                return f;
            }
            return specials[r % (sizeof(specials) / sizeof(specials[0]))];
        };
        a[i] = draw();
        b[i] = draw();
    }
    float *da, *db, *dfb;
    unsigned long long *dbad;
    hipMalloc(&da, a.size() * 4);
    hipMalloc(&db, b.size() * 4);
    hipMalloc(&dbad, 8);
    hipMalloc(&dfb, 16);
    hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
    hipMemset(dbad, 0, 8);
    hipMemset(dfb, 0, 16);
    hipLaunchKernelGGL(exact_kernel, dim3(1024), dim3(64), 0, 0, da, db, groups, dbad, dfb);
    unsigned long long bad = 0;
    float fb[4];
    hipMemcpy(&bad, dbad, 8, hipMemcpyDeviceToHost);
    hipMemcpy(fb, dfb, 16, hipMemcpyDeviceToHost);
    static const char *fn[] = {"U[-100,100)", "any bit pattern", "near-equal (cancellation)", "specials (0, -0, subnormal, inf, nan)"};
    printf("exact: mfma_4x4x1(A=x_j bcast, B=1, C=-x_i) vs v_sub_f32, %-38s pairs=%.3e mismatches=%llu", fn[flavour],
           (double)groups * 64 * 64, bad);
    if (bad) printf("  first: xj=%a xi=%a want=%a got=%a", fb[0], fb[1], fb[2], fb[3]);
    printf("\n");
    fflush(stdout);
    hipFree(da);
    hipFree(db);
    hipFree(dbad);
    hipFree(dfb);
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    for (int f = 0; f < 4; ++f) exactness(1u << 16, f);
    const int wps[] = {1, 2, 4};
    for (int w : wps) {
        run<16, 0, S_NONE>(w, iters);
        run<0, 8, S_4x4x1>(w, iters);
        run<0, 4, S_16x16x4>(w, iters);
        run<0, 2, S_32x32x2>(w, iters);
        run<16, 1, S_4x4x1>(w, iters);
        run<16, 2, S_4x4x1>(w, iters);
        run<16, 4, S_4x4x1>(w, iters);
        run<16, 8, S_4x4x1>(w, iters);
        run<32, 4, S_4x4x1>(w, iters);
        run<16, 1, S_16x16x4>(w, iters);
        run<32, 1, S_16x16x4>(w, iters);
        run<32, 2, S_16x16x4>(w, iters);
        run<32, 1, S_32x32x2>(w, iters);
        run<64, 1, S_32x32x2>(w, iters);
        run<64, 2, S_32x32x2>(w, iters);
    }
    return 0;
}
