// Micro-probe (not part of the product): sustained global_load_lds (16 B per lane) rate per CU from an L2-resident source,
// with 1 / 2 / 4 / 8 waves per CU issuing, and ds_read_b128 streaming alone for comparison.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void glds_kernel(const _Float16* src, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[72 * 1024 / 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const _Float16* p = src + ((size_t)blockIdx.x * 4096 + threadIdx.x) * 8;      // 64 KB window per block, L2 resident
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + (size_t)((it * 8 + i) & 7) * WAVES * 64 * 8),
                                             (__attribute__((address_space(3))) void*)(lds + (wave_u * 8 + i) * 512), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[blockIdx.x * WAVES * 64 + threadIdx.x] = (float)lds[threadIdx.x];
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void ldsread_kernel(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[64 * 1024 / 2];
    for (int i = threadIdx.x; i < 32 * 1024; i += WAVES * 64) lds[i] = (_Float16)(float)(i & 7);
    __syncthreads();
    h8 acc = {0, 0, 0, 0, 0, 0, 0, 0};
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            h8 v = *reinterpret_cast<const h8*>(&lds[((i * 64 + lane) * 8 + (threadIdx.x >> 6) * 8192) & (32 * 1024 - 8)]);
            acc += v;
        }
    }
    out[blockIdx.x * WAVES * 64 + threadIdx.x] = (float)acc[0];
}

template <int W> void run(const _Float16* src, float* out) {
    const int blocks = 256, iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(glds_kernel<W>, dim3(blocks), dim3(W * 64), 0, 0, src, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double bytes = (double)blocks * W * 64 * 16 * 8 * iters;
    printf("glds  %d waves/CU: %.3f ms  %.1f GB/s per CU  (%.0f GB/s chip)\n", W, ms, bytes / blocks / ms / 1e6, bytes / ms / 1e6);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(ldsread_kernel<W>, dim3(blocks), dim3(W * 64), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    hipEventElapsedTime(&ms, e0, e1);
    bytes = (double)blocks * W * 64 * 16 * 16 * iters;
    printf("ds_read_b128 %d waves/CU: %.3f ms  %.1f GB/s per CU\n", W, ms, bytes / blocks / ms / 1e6);
}

int main() {
    _Float16* src; float* out;
    hipMalloc(&src, 256ull * 4096 * 8 * 2 * 8); hipMemset(src, 0, 256ull * 4096 * 8 * 2 * 8);
    hipMalloc(&out, 256 * 512 * 4);
    run<1>(src, out); run<2>(src, out); run<4>(src, out); run<8>(src, out);
    return 0;
}
