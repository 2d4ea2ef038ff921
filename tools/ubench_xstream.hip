// ubench_xstream.hip -- what it costs to hand work to a second stream and wait for it on the first (the overlapped exchanges of a
// multi-GPU step: DESIGN.md section 5).  A step is  A (a kernel of ~ta us on stream 1) -> X (a tiny kernel, "the exchange") ->
// C (a kernel of ~tc us that does not need X) -> B (a kernel that needs X), timed over many steps with one wait at the end:
//   mode 0  everything on stream 1, in order:                         A X C B
//   mode 1  X on stream 2 between two events (hipEventRecord / hipStreamWaitEvent), C beside it:
//               s1: A rec(e1) C wait(e2) B        s2: wait(e1) X rec(e2)
//   mode 2  the same with hipStreamWriteValue32 / hipStreamWaitValue32 on two device words instead of events
//   mode 3  as mode 1 without C (the hand-off alone)
// The difference between mode 1 (or 2) and mode 0 MINUS what C could hide (min(tc, tx)) is the price of the two cross-stream waits.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ void spin(float *buf, int k)
{
    float x = buf[blockIdx.x * blockDim.x + threadIdx.x];
    for (int i = 0; i < k; ++i) x = x * 1.0001f + 0.5f;
    buf[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

#define CK(call)                                                                     \
    do {                                                                             \
        hipError_t e_ = (call);                                                      \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_));        \
            return 1;                                                                \
        }                                                                            \
    } while (0)

int main(int argc, char **argv)
{
    const int steps = argc > 1 ? atoi(argv[1]) : 400;
    const int blocks = 1024;  // fills the chip once over
    float *buf, *small;
    CK(hipMalloc((void **)&buf, (size_t)blocks * 256 * sizeof(float)));
    CK(hipMalloc((void **)&small, 256 * sizeof(float)));
    CK(hipMemset(buf, 0, (size_t)blocks * 256 * sizeof(float)));
    CK(hipMemset(small, 0, 256 * sizeof(float)));
    uint32_t *words;
    CK(hipMalloc((void **)&words, 256));
    CK(hipMemset(words, 0, 256));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e1, e2;
    CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
    // k iterations ~ microseconds: calibrate A (about 200 us) and C (about 25 us)
    auto time_kernel = [&](int k) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s1, buf, k);
        hipStreamSynchronize(s1);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s1, buf, k);
        hipStreamSynchronize(s1);
        return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 20.0;
    };
    int ka = 20000;
    const double per_iter = time_kernel(ka) / ka;
    ka = (int)(200.0 / per_iter);
    const int kc = (int)(25.0 / per_iter), kb = (int)(50.0 / per_iter);
    printf("kernels: A %.1f us, C %.1f us, B %.1f us (one workgroup per wave slot and more: %d workgroups); X: one workgroup, ~2 us\n",
           time_kernel(ka), time_kernel(kc), time_kernel(kb), blocks);
    for (int mode = 0; mode < 4; ++mode) {
        double best = 1e30;
        for (int rep = 0; rep < 3; ++rep) {
            uint32_t seq = 0;
            CK(hipMemset(words, 0, 256));
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            for (int st = 0; st < steps; ++st) {
                ++seq;
                hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s1, buf, ka);  // A
                if (mode == 0) {
                    hipLaunchKernelGGL(spin, dim3(1), dim3(256), 0, s1, small, 100);     // X
                    hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s1, buf, kc);   // C
                } else if (mode == 1 || mode == 3) {
                    CK(hipEventRecord(e1, s1));
                    CK(hipStreamWaitEvent(s2, e1, 0));
                    hipLaunchKernelGGL(spin, dim3(1), dim3(256), 0, s2, small, 100);     // X
                    CK(hipEventRecord(e2, s2));
                    if (mode == 1) hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s1, buf, kc);  // C
                    CK(hipStreamWaitEvent(s1, e2, 0));
                } else {
                    CK(hipStreamWriteValue32(s1, words, seq, 0));
                    CK(hipStreamWaitValue32(s2, words, seq, hipStreamWaitValueGte, 0xffffffffu));
                    hipLaunchKernelGGL(spin, dim3(1), dim3(256), 0, s2, small, 100);     // X
                    CK(hipStreamWriteValue32(s2, words + 16, seq, 0));
                    hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s1, buf, kc);   // C
                    CK(hipStreamWaitValue32(s1, words + 16, seq, hipStreamWaitValueGte, 0xffffffffu));
                }
                hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s1, buf, kb);  // B
            }
            CK(hipStreamSynchronize(s1));
            CK(hipStreamSynchronize(s2));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / steps;
            if (us < best) best = us;
        }
        printf("mode %d  %-62s %8.1f us per step\n", mode,
               mode == 0   ? "A X C B in order on one stream"
               : mode == 1 ? "X on a second stream between two events, C beside it"
               : mode == 2 ? "the same with hipStreamWriteValue32 / hipStreamWaitValue32"
                           : "X on a second stream between two events, no C (the hand-off alone)",
               best);
    }
    return 0;
}
