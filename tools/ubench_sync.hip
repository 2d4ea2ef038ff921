// ubench_sync.hip -- how long a host waits for a small kernel: hipStreamSynchronize against polling a word the kernel's last
// instruction writes into pinned host memory (the drop-in calls at the reference's own sizes are all latency: INTEGRATION.md
// section 6).  Prints the per-call wall time of launch + wait for both, for a kernel of K dependent steps.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ void work(float *buf, int k, volatile unsigned *done, unsigned seq)
{
    float x = buf[threadIdx.x];
    for (int i = 0; i < k; ++i) x = x * 1.0001f + 0.5f;
    buf[threadIdx.x] = x;
    if (done) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) *done = seq;
    }
}

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 2000;
    float *buf;
    unsigned *flag, *flag_dev;
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipHostMalloc((void **)&buf, 256 * sizeof(float), hipHostMallocMapped);
    hipHostMalloc((void **)&flag, 64, hipHostMallocMapped);
    hipHostGetDevicePointer((void **)&flag_dev, flag, 0);
    float *buf_dev;
    hipHostGetDevicePointer((void **)&buf_dev, buf, 0);
    for (int i = 0; i < 256; ++i) buf[i] = 1.f;
    *flag = 0;
    for (int k : {1, 1000, 4000}) {
        for (int mode = 0; mode < 4; ++mode) {
            unsigned seq = *flag;
            double best = 1e9, sum = 0;
            for (int r = 0; r < reps + 100; ++r) {
                auto t0 = std::chrono::steady_clock::now();
                ++seq;
                hipLaunchKernelGGL(work, dim3(1), dim3(256), 0, s, buf_dev, k, (mode == 1 || mode == 2) ? flag_dev : nullptr, seq);
                if (mode == 3) hipStreamWriteValue32(s, flag_dev, seq, 0);
                if (mode == 0) {
                    hipStreamSynchronize(s);
                } else if (mode == 1 || mode == 3) {
                    while (*(volatile unsigned *)flag != seq) {
                    }
                } else {  // poll, then let the runtime see the stream idle too (what a cautious host would do)
                    while (*(volatile unsigned *)flag != seq) {
                    }
                    hipStreamQuery(s);
                }
                auto t1 = std::chrono::steady_clock::now();
                const double us = std::chrono::duration<double, std::micro>(t1 - t0).count();
                if (r >= 100) {
                    sum += us;
                    if (us < best) best = us;
                }
            }
            hipStreamSynchronize(s);
            printf("k=%5d  %-28s mean %7.2f us  best %7.2f us\n", k, mode == 0 ? "hipStreamSynchronize" : mode == 1 ? "poll pinned word" : mode == 2 ? "poll + hipStreamQuery" : "hipStreamWriteValue32 + poll", sum / reps, best);
        }
    }
    return 0;
}
