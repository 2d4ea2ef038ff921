// ubench_dpp.hip -- does gfx950 execute the whole-wave DPP rotate (wave_rol:1), what does it cost, and which way does it turn?
// (diagnostic tool, not product code; behind the pair-symmetric FAST fold)
//
//   1. semantics: v_mov_b32_dpp dst, src wave_rol:1 -- lane l receives the value of lane (l + 1) % 64 ?
//   2. throughput: a stream of v_sub_f32_dpp acc, acc(wave_rol:1), x  against plain v_sub_f32, per waves per SIMD
//   3. the same for ds_bpermute_b32 (the LDS crossbar) as the alternative carrier
// Build: hipcc -O2 --offload-arch=gfx950 -o build/ubench_dpp tools/ubench_dpp.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <algorithm>
#include <vector>

__global__ void semantics(const float *in, float *out_rol, float *out_ror, float *out_bperm)
{
    const int l = threadIdx.x;
    const float v = in[l];
    out_rol[l] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x134, 0xf, 0xf, false));  // wave_rol:1
    out_ror[l] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x13c, 0xf, 0xf, false));  // wave_ror:1
    out_bperm[l] = __int_as_float(__builtin_amdgcn_ds_bpermute(((l + 1) & 63) * 4, __float_as_int(v)));
}

template <int KIND>  // 0 plain v_sub, 1 v_sub_dpp wave_rol:1, 2 ds_bpermute + v_sub, 3 v_sub_dpp row_shl:1
__global__ __launch_bounds__(256) void stream(uint64_t *out, float *sink, int iters)
{
    const int l = threadIdx.x & 63;
    float a0 = l, a1 = l + 1.f, a2 = l + 2.f, a3 = l + 3.f, a4 = l + 4.f, a5 = l + 5.f, a6 = l + 6.f, a7 = l + 7.f;
    float x = 1e-3f * l;
    const int addr = ((l + 1) & 63) * 4;
    uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
            if (KIND == 0)
                asm volatile("v_sub_f32 %0, %0, %8\n\tv_sub_f32 %1, %1, %8\n\tv_sub_f32 %2, %2, %8\n\tv_sub_f32 %3, %3, %8\n\t"
                             "v_sub_f32 %4, %4, %8\n\tv_sub_f32 %5, %5, %8\n\tv_sub_f32 %6, %6, %8\n\tv_sub_f32 %7, %7, %8\n\t"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));
            if (KIND == 1)
                asm volatile("v_sub_f32_dpp %0, %0, %8 wave_rol:1 row_mask:0xf bank_mask:0xf\n\tv_sub_f32_dpp %1, %1, %8 wave_rol:1 row_mask:0xf bank_mask:0xf\n\t"
                             "v_sub_f32_dpp %2, %2, %8 wave_rol:1 row_mask:0xf bank_mask:0xf\n\tv_sub_f32_dpp %3, %3, %8 wave_rol:1 row_mask:0xf bank_mask:0xf\n\t"
                             "v_sub_f32_dpp %4, %4, %8 wave_rol:1 row_mask:0xf bank_mask:0xf\n\tv_sub_f32_dpp %5, %5, %8 wave_rol:1 row_mask:0xf bank_mask:0xf\n\t"
                             "v_sub_f32_dpp %6, %6, %8 wave_rol:1 row_mask:0xf bank_mask:0xf\n\tv_sub_f32_dpp %7, %7, %8 wave_rol:1 row_mask:0xf bank_mask:0xf\n\t"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));
            if (KIND == 3)
                asm volatile("v_sub_f32_dpp %0, %0, %8 row_shl:1 row_mask:0xf bank_mask:0xf\n\tv_sub_f32_dpp %1, %1, %8 row_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                             "v_sub_f32_dpp %2, %2, %8 row_shl:1 row_mask:0xf bank_mask:0xf\n\tv_sub_f32_dpp %3, %3, %8 row_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                             "v_sub_f32_dpp %4, %4, %8 row_shl:1 row_mask:0xf bank_mask:0xf\n\tv_sub_f32_dpp %5, %5, %8 row_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                             "v_sub_f32_dpp %6, %6, %8 row_shl:1 row_mask:0xf bank_mask:0xf\n\tv_sub_f32_dpp %7, %7, %8 row_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));
            if (KIND == 2) {
                a0 = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(a0))) - x;
                a1 = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(a1))) - x;
                a2 = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(a2))) - x;
                a3 = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(a3))) - x;
                a4 = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(a4))) - x;
                a5 = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(a5))) - x;
                a6 = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(a6))) - x;
                a7 = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(a7))) - x;
            }
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (l == 0) {
        out[2 * wave] = t1 - t0;
        out[2 * wave + 1] = r1 - r0;
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 123.456f) sink[0] = a0;
}

template <int KIND>
static void run(int wps, int iters)
{
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * wps, waves = blocks * 4;
    uint64_t *d;
    float *sink;
    (void)hipMalloc(&d, sizeof(uint64_t) * 2 * waves);
    (void)hipMalloc(&sink, 64);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(stream<KIND>, dim3(blocks), dim3(256), 0, 0, d, sink, iters / 10 + 1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(stream<KIND>, dim3(blocks), dim3(256), 0, 0, d, sink, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(2 * waves);
    (void)hipMemcpy(h.data(), d, sizeof(uint64_t) * 2 * waves, hipMemcpyDeviceToHost);
    std::vector<double> clk(waves);
    for (int w = 0; w < waves; ++w) clk[w] = (double)h[2 * w] / (double)h[2 * w + 1] * 100.0;
    std::sort(clk.begin(), clk.end());
    const double ninst = (double)iters * 64.0;
    static const char *nm[] = {"v_sub_f32", "v_sub_f32_dpp wave_rol:1", "ds_bpermute_b32 + v_sub_f32", "v_sub_f32_dpp row_shl:1"};
    printf("%-30s waves/SIMD=%d  wall-cyc per (rotate+sub) per SIMD = %6.2f  clock=%5.0f MHz\n", nm[KIND], wps,
           (ms * 1e-3) * (clk[waves / 2] * 1e6) / ninst / wps, clk[waves / 2]);
    fflush(stdout);
    (void)hipFree(d);
    (void)hipFree(sink);
}

int main()
{
    float h[64], *din, *d1, *d2, *d3, o1[64], o2[64], o3[64];
    for (int i = 0; i < 64; ++i) h[i] = (float)i;
    (void)hipMalloc(&din, 256);
    (void)hipMalloc(&d1, 256);
    (void)hipMalloc(&d2, 256);
    (void)hipMalloc(&d3, 256);
    (void)hipMemcpy(din, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(semantics, dim3(1), dim3(64), 0, 0, din, d1, d2, d3);
    (void)hipMemcpy(o1, d1, 256, hipMemcpyDeviceToHost);
    (void)hipMemcpy(o2, d2, 256, hipMemcpyDeviceToHost);
    (void)hipMemcpy(o3, d3, 256, hipMemcpyDeviceToHost);
    int ok_rol = 1, ok_ror = 1, ok_bp = 1;
    for (int l = 0; l < 64; ++l) {
        ok_rol &= o1[l] == (float)((l + 1) & 63);
        ok_ror &= o2[l] == (float)((l + 63) & 63);
        ok_bp &= o3[l] == (float)((l + 1) & 63);
    }
    printf("wave_rol:1: lane l <- lane (l+1)%%64: %s   [lanes 0,1,15,16,62,63 got %g %g %g %g %g %g]\n", ok_rol ? "yes" : "NO", o1[0], o1[1],
           o1[15], o1[16], o1[62], o1[63]);
    printf("wave_ror:1: lane l <- lane (l-1)%%64: %s   [lanes 0,1,15,16,62,63 got %g %g %g %g %g %g]\n", ok_ror ? "yes" : "NO", o2[0], o2[1],
           o2[15], o2[16], o2[62], o2[63]);
    printf("ds_bpermute (l+1)%%64: %s\n", ok_bp ? "yes" : "NO");
    for (int w : {1, 2, 4, 8}) {
        run<0>(w, 4000);
        run<1>(w, 4000);
        run<3>(w, 4000);
        run<2>(w, 4000);
    }
    return 0;
}
