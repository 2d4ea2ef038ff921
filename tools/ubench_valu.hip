// ubench_valu.hip -- VALU issue-rate microbenchmark for gfx950 (diagnostic tool, not product code).
//
// Question it answers, for the pair-fold kernels' design: how many cycles does a wave64 spend per
// v_fma_f32 / v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 / v_rcp_f32 / v_div_scale_f32 as a function of the number of
// waves per SIMD, and what mix rate does the FAST pair body (9 full-rate ops + 1 rcp) reach?
//
// Build: hipcc -O2 --offload-arch=gfx950 -o build/ubench_valu tools/ubench_valu.hip
// Run:   build/ubench_valu            (prints one line per kind x waves/SIMD)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum Kind { K_FMA = 0, K_PKFMA, K_PKMUL, K_PKADD, K_RCP, K_MUL, K_ADD, K_DIVSCALE, K_MIXFAST, K_MIXFAST_PK, K_FMA_SGPR, K_PKMUL_SGPR, K_FMA_INLINE, K_ADD_LIT, K_NKINDS };
static const char *kind_name[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_rcp_f32", "v_mul_f32",
                                  "v_add_f32", "v_div_scale_f32", "mix:9fma+1rcp", "mix:pk(2 pairs)=9pk+2rcp", "v_fma_f32 (SGPR src1)",
                                  "v_pk_mul_f32 (SGPR pair)", "v_fma_f32 (inline 1.0)", "v_add_f32 (literal)"};
// VALU instructions per loop iteration and "lane-ops" (scalar-equivalent ops) per instruction
static const int insts_per_iter[] = {8, 8, 8, 8, 8, 8, 8, 8, 10, 11, 8, 8, 8, 8};

template <int KIND>
__global__ __launch_bounds__(256) void ub(uint64_t *out, int iters, float seed)
{
    const float l = (float)(threadIdx.x & 63) * 1e-3f;
    float a0 = seed + l, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f,
          a7 = a0 + 7.f;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4},
       p7 = {a7, a6};
    const float b = 1.0000001f, c = 1e-9f;
    const f2 pb = {b, b}, pc = {c, c};
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int rep = 0; rep < 8; ++rep) {
        if (KIND == K_FMA) {
#define X(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n\t"
            asm volatile(REP8(X) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
#undef X
        } else if (KIND == K_FMA_SGPR) {  // the same stream with the multiplier in a scalar register
            const float sb = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(b)));
#define X(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n\t"
            asm volatile(REP8(X) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sb), "v"(c));
#undef X
        } else if (KIND == K_PKMUL_SGPR) {
            const float sb1 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(b)));
            f2 spb = {sb1, sb1};
#define X(n) "v_pk_mul_f32 %" #n ", %" #n ", %8\n\t"
            asm volatile(REP8(X) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "s"(spb), "v"(pc));
#undef X
        } else if (KIND == K_FMA_INLINE) {  // an inline constant as the addend
#define X(n) "v_fma_f32 %" #n ", %" #n ", %8, 1.0\n\t"
            asm volatile(REP8(X) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(c));
#undef X
        } else if (KIND == K_ADD_LIT) {  // a 32-bit literal operand
#define X(n) "v_add_f32 %" #n ", 0x3a83126f, %" #n "\n\t"
            asm volatile(REP8(X) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
#undef X
        } else if (KIND == K_MUL) {
#define X(n) "v_mul_f32 %" #n ", %" #n ", %8\n\t"
            asm volatile(REP8(X) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
#undef X
        } else if (KIND == K_ADD) {
#define X(n) "v_add_f32 %" #n ", %" #n ", %9\n\t"
            asm volatile(REP8(X) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
#undef X
        } else if (KIND == K_RCP) {
#define X(n) "v_rcp_f32 %" #n ", %" #n "\n\t"
            asm volatile(REP8(X) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
#undef X
        } else if (KIND == K_DIVSCALE) {
#define X(n) "v_div_scale_f32 %" #n ", vcc, %" #n ", %8, %" #n "\n\t"
            asm volatile(REP8(X) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
#undef X
        } else if (KIND == K_PKFMA) {
#define X(n) "v_pk_fma_f32 %" #n ", %" #n ", %8, %9\n\t"
            asm volatile(REP8(X) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));
#undef X
        } else if (KIND == K_PKMUL) {
#define X(n) "v_pk_mul_f32 %" #n ", %" #n ", %8\n\t"
            asm volatile(REP8(X) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));
#undef X
        } else if (KIND == K_PKADD) {
#define X(n) "v_pk_add_f32 %" #n ", %" #n ", %9\n\t"
            asm volatile(REP8(X) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));
#undef X
        } else if (KIND == K_MIXFAST) {
            // one FAST pair: 3 sub, 3 fma (r2), 1 rcp, 3 fma (accumulate); a0..a2 = acc, a3..a5 = d, a6 = r2, a7 = inv
            asm volatile(
                "v_sub_f32 %3, %8, %3\n\t"
                "v_sub_f32 %4, %8, %4\n\t"
                "v_sub_f32 %5, %8, %5\n\t"
                "v_fma_f32 %6, %3, %3, %9\n\t"
                "v_fma_f32 %6, %4, %4, %6\n\t"
                "v_fma_f32 %6, %5, %5, %6\n\t"
                "v_rcp_f32 %7, %6\n\t"
                "v_fma_f32 %0, %3, %7, %0\n\t"
                "v_fma_f32 %1, %4, %7, %1\n\t"
                "v_fma_f32 %2, %5, %7, %2\n\t"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                : "v"(b), "v"(c));
        } else if (KIND == K_MIXFAST_PK) {
            // two FAST pairs (2 bodies x 1 j) with packed math: 3 pk_add, 3 pk_fma, 2 rcp, 3 pk_fma = 11 insts
            // p0..p2 = acc, p3..p5 = d, p6 = r2, p7 = inv; the rcps run on a6/a7 (throughput, not dataflow, is measured)
            asm volatile(
                "v_pk_add_f32 %3, %10, %3\n\t"
                "v_pk_add_f32 %4, %10, %4\n\t"
                "v_pk_add_f32 %5, %10, %5\n\t"
                "v_pk_fma_f32 %6, %3, %3, %11\n\t"
                "v_pk_fma_f32 %6, %4, %4, %6\n\t"
                "v_pk_fma_f32 %6, %5, %5, %6\n\t"
                "v_rcp_f32 %8, %8\n\t"
                "v_rcp_f32 %9, %9\n\t"
                "v_pk_fma_f32 %0, %3, %7, %0\n\t"
                "v_pk_fma_f32 %1, %4, %7, %1\n\t"
                "v_pk_fma_f32 %2, %5, %7, %2\n\t"
                : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7), "+v"(a6), "+v"(a7)
                : "v"(pb), "v"(pc));
        }
      }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint64_t r1 = __builtin_amdgcn_s_memrealtime();
    float sink = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x +
                 p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) {
        out[2 * wave] = t1 - t0;
        out[2 * wave + 1] = r1 - r0;
    }
    if (sink == 123.456f) out[0] = 0;  // keep the accumulators live
}

template <int KIND>
static void run(int waves_per_simd, int iters)
{
    int dev = 0;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, dev);
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * waves_per_simd;  // 256-thread blocks: 4 waves = one per SIMD
    const int waves = blocks * 4;
    uint64_t *d;
    hipMalloc(&d, sizeof(uint64_t) * 2 * waves);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(ub<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters / 10 + 1, 1.0f);  // warm
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(ub<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(2 * waves);
    hipMemcpy(h.data(), d, sizeof(uint64_t) * 2 * waves, hipMemcpyDeviceToHost);
    std::vector<double> cyc(waves), clk(waves);
    for (int w = 0; w < waves; ++w) {
        cyc[w] = (double)h[2 * w];
        clk[w] = (double)h[2 * w] / (double)h[2 * w + 1] * 100.0;  // MHz: s_memrealtime ticks at 100 MHz
    }
    std::sort(cyc.begin(), cyc.end());
    std::sort(clk.begin(), clk.end());
    const double ninst = (double)iters * 8.0 * insts_per_iter[KIND];
    const double cyc_per_inst_wave = cyc[waves / 2] / ninst;                 // one wave's view
    const double cyc_per_inst_simd = cyc_per_inst_wave / waves_per_simd;    // SIMD throughput view
    const double wall_inst_rate = ninst * waves / (ms * 1e-3);               // wave-instructions / s, chip
    const double wall_cyc_per_inst_simd = (ms * 1e-3) * (clk[waves / 2] * 1e6) / ninst / waves_per_simd;
    printf("[cus=%d] ", cus);
    printf("%-20s waves/SIMD=%d  cyc/inst(one wave)=%6.2f  cyc/inst(SIMD)=%5.2f  clock=%5.0f MHz  wall=%.3f ms  "
           "chip wave-inst/s=%.3e  wall-cyc/inst(SIMD)=%.2f\n",
           kind_name[KIND], waves_per_simd, cyc_per_inst_wave, cyc_per_inst_simd, clk[waves / 2], ms, wall_inst_rate, wall_cyc_per_inst_simd);
    hipFree(d);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    const int wps[] = {1, 2, 4, 8};
    for (int w : wps) {
        run<K_FMA>(w, iters);
        run<K_FMA_SGPR>(w, iters);
        run<K_FMA_INLINE>(w, iters);
        run<K_ADD_LIT>(w, iters);
        run<K_PKMUL_SGPR>(w, iters);
        run<K_MUL>(w, iters);
        run<K_ADD>(w, iters);
        run<K_PKFMA>(w, iters);
        run<K_PKMUL>(w, iters);
        run<K_PKADD>(w, iters);
        run<K_RCP>(w, iters);
        run<K_DIVSCALE>(w, iters);
        run<K_MIXFAST>(w, iters);
        run<K_MIXFAST_PK>(w, iters);
    }
    return 0;
}
