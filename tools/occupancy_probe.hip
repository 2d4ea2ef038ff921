// occupancy_probe.hip -- census of workgroup residency per CU (diagnostic tool).
// Launches a kernel of T threads and L bytes of LDS per block that spins for a fixed time and records where and when it
// ran; the host then counts, per CU, the maximum number of blocks alive at once.
// Build: hipcc -O2 --offload-arch=gfx950 -o build/occupancy_probe tools/occupancy_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <vector>

struct Rec {
    uint32_t hwid, xcc;
    uint64_t t0, t1;
};

__global__ void probe(Rec *out, unsigned long long spin_ticks)
{
    extern __shared__ char lds[];
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    lds[threadIdx.x] = (char)threadIdx.x;
    __syncthreads();
    while (__builtin_amdgcn_s_memrealtime() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    if (threadIdx.x == 0) {
        Rec r;
        r.hwid = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, all 32 bits
        r.xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
        r.t0 = t0;
        r.t1 = __builtin_amdgcn_s_memrealtime();
        r.hwid ^= (uint32_t)lds[5] & 0u;
        out[blockIdx.x] = r;
    }
}

int main(int argc, char **argv)
{
    const int threads = argc > 1 ? atoi(argv[1]) : 576;
    const int lds = argc > 2 ? atoi(argv[2]) : 57376;
    const int blocks = argc > 3 ? atoi(argv[3]) : 2048;
    Rec *d;
    hipMalloc(&d, sizeof(Rec) * blocks);
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    int api = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, probe, threads, lds);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), lds, 0, d, 5000ull);  // 50 us at 100 MHz
    hipError_t e = hipDeviceSynchronize();
    std::vector<Rec> h(blocks);
    hipMemcpy(h.data(), d, sizeof(Rec) * blocks, hipMemcpyDeviceToHost);
    std::map<uint32_t, std::vector<std::pair<uint64_t, int>>> ev;
    for (auto &r : h) {
        const uint32_t cu = (r.hwid >> 8) & 0xf, sh = (r.hwid >> 12) & 1, se = (r.hwid >> 13) & 7;
        const uint32_t key = ((r.xcc & 0xf) << 16) | (se << 8) | (sh << 4) | cu;
        ev[key].push_back({r.t0, +1});
        ev[key].push_back({r.t1, -1});
    }
    int worst = 0, best = 1 << 30;
    for (auto &kv : ev) {
        auto &v = kv.second;
        std::sort(v.begin(), v.end());
        int cur = 0, mx = 0;
        for (auto &p : v) {
            cur += p.second;
            mx = std::max(mx, cur);
        }
        worst = std::max(worst, mx);
        best = std::min(best, mx);
    }
    printf("threads=%d lds=%d blocks=%d: occupancy API says %d blocks/CU; census: %zu distinct CUs, max concurrent blocks per CU "
           "min=%d max=%d (%s)\n",
           threads, lds, blocks, api, ev.size(), best, worst, hipGetErrorString(e));
    return 0;
}
