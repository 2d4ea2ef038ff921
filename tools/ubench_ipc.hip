// ubench_ipc.hip -- can two PROCESSES order their streams through a word in each other's device memory, and read each other's
// buffers, without a collective library?  (The exchange of a multi-GPU step as pulls over xGMI: DESIGN.md section 5.)
// Parent and child (fork before any HIP call) each allocate a data buffer and a flag word, swap hipIpcMemHandle_t's through a pipe,
// then run ROUNDS rounds of:   fill my buffer with (rank, round) [kernel]  ->  hipStreamWriteValue32(my flag = round)  ->
//   hipStreamWaitValue32(PEER's flag >= round, through the IPC mapping)  ->  pull the peer's buffer into a local one [kernel]  -> check.
// No host synchronisation between the processes inside the loop: everything is ordered on the streams.  Prints the verdict and the
// time per round.  Both processes share whatever GPU is visible (one, on the build's boxes).
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(call)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "[%d] %s failed: %s\n", rank, #call, hipGetErrorString(e_));           \
            _exit(10);                                                                             \
        }                                                                                          \
    } while (0)

__global__ void fill(float4 *buf, uint32_t n, float rank, float round)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) buf[i] = make_float4((float)i, rank, round, -(float)i);
}
__global__ void pull(const float4 *__restrict__ src, float4 *__restrict__ dst, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
__global__ void check(const float4 *buf, uint32_t n, float rank, float round, uint32_t *bad)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 r = buf[i];
    if (!(r.x == (float)i && r.y == rank && r.z == round && r.w == -(float)i)) atomicAdd(bad, 1u);
}

struct Handles {
    hipIpcMemHandle_t data, flag;
};

static bool xfer(int fd, void *p, size_t n, bool wr)
{
    char *c = (char *)p;
    while (n) {
        ssize_t k = wr ? write(fd, c, n) : read(fd, c, n);
        if (k <= 0) return false;
        c += k, n -= (size_t)k;
    }
    return true;
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 200;
    const uint32_t n = 16384;  // records: 256 KB, one rank's slot of positions at 8 ranks
    int to_child[2], to_parent[2];
    if (pipe(to_child) || pipe(to_parent)) return 2;
    const pid_t pid = fork();
    const int rank = pid == 0 ? 1 : 0;
    const int rfd = rank ? to_child[0] : to_parent[0], wfd = rank ? to_parent[1] : to_child[1];

    float4 *data, *local;
    uint32_t *flag, *bad;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipMalloc((void **)&data, n * sizeof(float4)));
    CK(hipMalloc((void **)&local, n * sizeof(float4)));
    CK(hipMalloc((void **)&flag, 4096));
    CK(hipMalloc((void **)&bad, 64));
    CK(hipMemset(flag, 0, 4096));
    CK(hipMemset(bad, 0, 64));
    CK(hipDeviceSynchronize());
    Handles mine, theirs;
    CK(hipIpcGetMemHandle(&mine.data, data));
    CK(hipIpcGetMemHandle(&mine.flag, flag));
    if (!xfer(wfd, &mine, sizeof(mine), true) || !xfer(rfd, &theirs, sizeof(theirs), false)) {
        fprintf(stderr, "[%d] handle exchange failed\n", rank);
        _exit(11);
    }
    float4 *peer_data;
    uint32_t *peer_flag;
    CK(hipIpcOpenMemHandle((void **)&peer_data, theirs.data, hipIpcMemLazyEnablePeerAccess));
    CK(hipIpcOpenMemHandle((void **)&peer_flag, theirs.flag, hipIpcMemLazyEnablePeerAccess));

    const dim3 grid((n + 255) / 256), block(256);
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 1; r <= rounds; ++r) {
        hipLaunchKernelGGL(fill, grid, block, 0, s, data, n, (float)rank, (float)r);
        CK(hipStreamWriteValue32(s, flag, (uint32_t)r, 0));
        CK(hipStreamWaitValue32(s, peer_flag, (uint32_t)r, hipStreamWaitValueGte, 0xffffffffu));
        hipLaunchKernelGGL(pull, grid, block, 0, s, (const float4 *)peer_data, local, n);
        hipLaunchKernelGGL(check, grid, block, 0, s, (const float4 *)local, n, (float)(1 - rank), (float)r, bad);
        // (the peer must not refill its buffer for round r + 1 before this pull is through: a second word, the other way round)
        CK(hipStreamWriteValue32(s, flag + 16, (uint32_t)r, 0));
        CK(hipStreamWaitValue32(s, peer_flag + 16, (uint32_t)r, hipStreamWaitValueGte, 0xffffffffu));
    }
    CK(hipStreamSynchronize(s));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
    uint32_t nbad = 0;
    CK(hipMemcpy(&nbad, bad, sizeof(nbad), hipMemcpyDeviceToHost));
    printf("[rank %d] %d rounds of fill -> signal -> wait(peer) -> pull 256 KB -> check -> signal -> wait: %u bad records, %.1f us per round\n", rank, rounds,
           nbad, us);
    fflush(stdout);
    CK(hipIpcCloseMemHandle(peer_data));
    CK(hipIpcCloseMemHandle(peer_flag));
    if (rank == 1) _exit(nbad ? 1 : 0);
    int status = 0;
    waitpid(pid, &status, 0);
    const int child = WIFEXITED(status) ? WEXITSTATUS(status) : 99;
    printf("verdict: %s (child exit %d)\n", (nbad == 0 && child == 0) ? "OK" : "FAILED", child);
    return (nbad == 0 && child == 0) ? 0 : 1;
}
