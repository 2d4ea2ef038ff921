// ladder_exhaustive.hip -- exhaustive check of shared-reciprocal division ladders against the IEEE binary32 divide.
//
// n / d in binary32 with every intermediate normal depends on the two significands only: multiplication, FMA and
// rounding commute with scaling by powers of two, and so does v_rcp_f32 (checked here for every significand over the
// exponents the STRICT range guard admits, `rcp` mode).  So comparing a ladder with '/' for ALL 2^23 x 2^23 pairs of
// significands (d in [1,2), n in [1,2): quotients in (1/2, 2)) settles it for every guarded input.
//
//   ladder_exhaustive [first_md [count_md]]     default: all 2^23 denominators
//
// Variants (r0 = v_rcp_f32(d); e = fma(-d, r0, 1); r = fma(e, r0, r0)):
//   full   q0 = n*r;  t0 = fma(-d,q0,n); q1 = fma(t0,r,q0);  t1 = fma(-d,q1,n); q = fma(t1,r,q1)     (what the compiler's divide does, minus scaling)
//   short  q0 = n*r;  t0 = fma(-d,q0,n); q  = fma(t0,r,q0)
//   raw    the full ladder on r0 instead of r (no reciprocal refinement)
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -o build/ladder_exhaustive tools/ladder_exhaustive.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            std::fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));       \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

struct Tally {
    unsigned long long bad[3];  // full, short, raw
    uint32_t first[3][2];       // one offending (n, d) per variant, as bit patterns
};

__device__ __forceinline__ float ladder(float n, float d, float r, bool two_steps)
{
    const float q0 = n * r;
    const float t0 = __builtin_fmaf(-d, q0, n);
    const float q1 = __builtin_fmaf(t0, r, q0);
    if (!two_steps) return q1;
    const float t1 = __builtin_fmaf(-d, q1, n);
    return __builtin_fmaf(t1, r, q1);
}

// one workgroup per denominator significand; its 256 threads stride over all 2^23 numerator significands
__global__ __launch_bounds__(256) void pairs_kernel(uint32_t first_md, Tally *tally)
{
    const uint32_t md = first_md + blockIdx.x;
    const float d = __uint_as_float(0x3f800000u | md);
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r0, 1.0f);
    const float r = __builtin_fmaf(e, r0, r0);
    uint32_t bad0 = 0, bad1 = 0, bad2 = 0, w0 = 0, w1 = 0, w2 = 0;
    for (uint32_t mn = threadIdx.x; mn < (1u << 23); mn += 256u) {
        const float n = __uint_as_float(0x3f800000u | mn);
        const uint32_t ref = __float_as_uint(n / d);
        const uint32_t a = __float_as_uint(ladder(n, d, r, true));
        const uint32_t b = __float_as_uint(ladder(n, d, r, false));
        const uint32_t c = __float_as_uint(ladder(n, d, r0, true));
        if (a != ref) { ++bad0; w0 = mn | 0x80000000u; }
        if (b != ref) { ++bad1; w1 = mn | 0x80000000u; }
        if (c != ref) { ++bad2; w2 = mn | 0x80000000u; }
    }
    const uint32_t bads[3] = {bad0, bad1, bad2}, ws[3] = {w0, w1, w2};
    for (int v = 0; v < 3; ++v)
        if (bads[v]) {
            atomicAdd(&tally->bad[v], (unsigned long long)bads[v]);
            tally->first[v][0] = 0x3f800000u | (ws[v] & 0x7fffffu);
            tally->first[v][1] = 0x3f800000u | md;
        }
}

// v_rcp_f32 commutes with scaling: rcp(m * 2^k) * 2^k == rcp(m), bit for bit, for every significand and k in [k_lo, k_hi]
__global__ __launch_bounds__(256) void rcp_scale_kernel(int k_lo, int k_hi, unsigned long long *bad)
{
    const uint32_t m = blockIdx.x * 256u + threadIdx.x;  // 2^23 threads
    const float base = __builtin_amdgcn_rcpf(__uint_as_float(0x3f800000u | m));
    uint32_t b = 0;
    for (int k = k_lo; k <= k_hi; ++k) {
        const float d = __uint_as_float(((uint32_t)(127 + k) << 23) | m);
        const float r = __builtin_amdgcn_rcpf(d);
        // r is 2^-k * base exactly iff r * 2^k == base (the product is exact: a power of two, no under/overflow in this range)
        const float back = r * __uint_as_float((uint32_t)(127 + k) << 23);
        if (__float_as_uint(back) != __float_as_uint(base)) ++b;
    }
    if (b) atomicAdd(bad, (unsigned long long)b);
}

int main(int argc, char **argv)
{
    const uint32_t first = argc > 1 ? (uint32_t)std::strtoul(argv[1], nullptr, 0) : 0u;
    const uint32_t total = argc > 2 ? (uint32_t)std::strtoul(argv[2], nullptr, 0) : (1u << 23) - first;
    Tally *tally = nullptr;
    unsigned long long *rbad = nullptr;
    CHECK(hipMalloc((void **)&tally, sizeof(Tally)));
    CHECK(hipMalloc((void **)&rbad, sizeof(*rbad)));
    CHECK(hipMemset(tally, 0, sizeof(Tally)));
    CHECK(hipMemset(rbad, 0, sizeof(*rbad)));

    // denominators the guard admits for the reference constants span 2^-24 .. 2^44; check a wider band
    hipLaunchKernelGGL(rcp_scale_kernel, dim3((1u << 23) / 256), dim3(256), 0, 0, -60, 60, rbad);
    CHECK(hipDeviceSynchronize());
    unsigned long long rb = 0;
    CHECK(hipMemcpy(&rb, rbad, sizeof(rb), hipMemcpyDeviceToHost));
    std::printf("v_rcp_f32 scale invariance, 2^23 significands x exponents -60..60: %llu violations\n", rb);
    std::fflush(stdout);

    const uint32_t slab = 1u << 13;  // denominators per launch: 2^36 pairs
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t done = 0;
    while (done < total) {
        const uint32_t cnt = total - done < slab ? total - done : slab;
        hipLaunchKernelGGL(pairs_kernel, dim3(cnt), dim3(256), 0, 0, first + done, tally);
        CHECK(hipDeviceSynchronize());
        done += cnt;
        if ((done / slab) % 32 == 0 || done == total) {
            Tally h;
            CHECK(hipMemcpy(&h, tally, sizeof(h), hipMemcpyDeviceToHost));
            const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::printf("denominators %u..%u of 2^23 done (%.3e pairs, %.0f s): mismatches full %llu, short %llu, raw-reciprocal %llu\n",
                        first, first + done, (double)done * 8388608.0, s, h.bad[0], h.bad[1], h.bad[2]);
            std::fflush(stdout);
        }
    }
    Tally h;
    CHECK(hipMemcpy(&h, tally, sizeof(h), hipMemcpyDeviceToHost));
    const char *names[3] = {"full", "short", "raw-reciprocal"};
    for (int v = 0; v < 3; ++v)
        if (h.bad[v]) {
            float n, d;
            std::memcpy(&n, &h.first[v][0], 4);
            std::memcpy(&d, &h.first[v][1], 4);
            std::printf("%s: %llu mismatches, e.g. n = %a (0x%08x), d = %a (0x%08x)\n", names[v], h.bad[v], n, h.first[v][0], d, h.first[v][1]);
        } else {
            std::printf("%s: exact for all %.3e pairs checked\n", names[v], (double)total * 8388608.0);
        }
    return 0;
}
