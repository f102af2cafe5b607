// Micro-probe (not part of the product): can two PROCESSES that share one GPU exchange data through IPC-mapped device memory
// while kernels of both are running?  (The frame-shard communicator of the library — csrc/comm.hip — relies on exactly that on a
// 1-GPU box, and on the same primitives over xGMI on an 8-GPU node.)
//   1. hipExtMallocWithFlags(fine-grained) + hipIpcGetMemHandle / hipIpcOpenMemHandle across a fork()
//   2. a ping-pong of system-scope flag stores / spinning loads between one-block kernels of the two processes: round-trip latency
//   3. a 16 MB peer write followed by a flag, checked word by word on the other side
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-result"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "[%d] %s:%d %s -> %s\n", me, __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                            \
        }                                                                                       \
    } while (0)

static int me = 0;

__global__ void pingpong(unsigned* mine, unsigned* peer, int first, int rounds, long long* cycles, int* fail) {
    const long long t0 = wall_clock64();
    for (int r = 1; r <= rounds; ++r) {
        if (first) __hip_atomic_store(peer, (unsigned)r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        long spins = 0;
        while (__hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned)r) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > 200000000L) { *fail = r; return; }
        }
        if (!first) __hip_atomic_store(peer, (unsigned)r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    *cycles = wall_clock64() - t0;
}

__global__ void push(unsigned* dst, long n, unsigned seed) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = seed + (unsigned)i * 2654435761u;
}
__global__ void set_flag(unsigned* flag, unsigned v) {
    __threadfence_system();
    __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void wait_flag(unsigned* flag, unsigned v, int* fail) {
    long spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < v) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > 100000000L) { *fail = 1; return; }
    }
}
__global__ void check(const unsigned* src, long n, unsigned seed, unsigned long long* bad) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        if (src[i] != seed + (unsigned)i * 2654435761u) atomicAdd(bad, 1ull);
}

int main(int argc, char** argv) {
    const int finegrained = argc > 1 ? atoi(argv[1]) : 1;
    int p2c[2], c2p[2];
    if (pipe(p2c) || pipe(c2p)) return 1;
    pid_t pid = fork();                       // before any HIP call
    me = pid == 0 ? 1 : 0;
    const int rfd = me ? p2c[0] : c2p[0], wfd = me ? c2p[1] : p2c[1];
    const long NW = 4l << 20;                 // 16 MB payload
    unsigned *flags, *inbox;
    if (finegrained) {
        CK(hipExtMallocWithFlags((void**)&flags, 4096, hipDeviceMallocFinegrained));
        CK(hipExtMallocWithFlags((void**)&inbox, NW * 4, hipDeviceMallocFinegrained));
    } else {
        CK(hipMalloc((void**)&flags, 4096));
        CK(hipMalloc((void**)&inbox, NW * 4));
    }
    CK(hipMemset(flags, 0, 4096));
    CK(hipMemset(inbox, 0xff, NW * 4));
    CK(hipDeviceSynchronize());
    hipIpcMemHandle_t h[2], ph[2];
    CK(hipIpcGetMemHandle(&h[0], flags));
    CK(hipIpcGetMemHandle(&h[1], inbox));
    if (write(wfd, h, sizeof(h)) != (ssize_t)sizeof(h) || read(rfd, ph, sizeof(ph)) != (ssize_t)sizeof(ph)) return 3;
    unsigned *pflags, *pinbox;
    CK(hipIpcOpenMemHandle((void**)&pflags, ph[0], hipIpcMemLazyEnablePeerAccess));
    CK(hipIpcOpenMemHandle((void**)&pinbox, ph[1], hipIpcMemLazyEnablePeerAccess));
    long long* cyc;
    int* fail;
    unsigned long long* bad;
    CK(hipHostMalloc((void**)&cyc, 8));
    CK(hipHostMalloc((void**)&fail, 4));
    CK(hipHostMalloc((void**)&bad, 8));
    *cyc = 0; *fail = 0; *bad = 0;
    char tok = 1;                              // both sides mapped: go
    if (write(wfd, &tok, 1) != 1 || read(rfd, &tok, 1) != 1) return 3;

    const int rounds = 2000;
    hipLaunchKernelGGL(pingpong, dim3(1), dim3(1), 0, 0, flags, pflags, me == 0, rounds, cyc, fail);
    CK(hipDeviceSynchronize());
    printf("[%d] ping-pong (%s memory): %s, %.2f us per round trip (wall_clock64 at 100 MHz)\n", me, finegrained ? "fine-grained" : "coarse-grained",
           *fail ? "GAVE UP" : "ok", *cyc / 100.0 / rounds);

    // payload: me writes into the peer's inbox, then raises the peer's flag word 1; peer waits and checks
    hipLaunchKernelGGL(push, dim3(512), dim3(256), 0, 0, pinbox, NW, 1000u + me);
    hipLaunchKernelGGL(set_flag, dim3(1), dim3(1), 0, 0, pflags + 1, 7u);
    hipLaunchKernelGGL(wait_flag, dim3(1), dim3(1), 0, 0, flags + 1, 7u, fail);
    hipLaunchKernelGGL(check, dim3(512), dim3(256), 0, 0, inbox, NW, 1000u + (1 - me), bad);
    CK(hipDeviceSynchronize());
    printf("[%d] 16 MB peer write + flag: %s, %llu wrong words\n", me, *fail ? "GAVE UP" : "ok", *bad);
    if (write(wfd, &tok, 1) != 1 || read(rfd, &tok, 1) != 1) return 3;      // nobody unmaps while the other still reads
    CK(hipIpcCloseMemHandle(pflags));
    CK(hipIpcCloseMemHandle(pinbox));
    if (me == 0) {
        int st = 0;
        waitpid(pid, &st, 0);
        return WEXITSTATUS(st);
    }
    return (*fail || *bad) ? 4 : 0;
}
