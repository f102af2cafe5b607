// Micro-probe (not part of the product): how fast can 8 waves / CU issue the 256x320 tile's MFMA stream when operands are
// already in registers?  Variants: 16x16x32 with the kernel's 10x4 accumulator tile, and 32x32x16 with a 5x2 tile.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(512, 2) void probe(float* out, const _Float16* in, int iters) {
    h8 a[10], b[4];
    for (int i = 0; i < 10; ++i) a[i] = *reinterpret_cast<const h8*>(in + (threadIdx.x + i * 512) * 8);
    for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const h8*>(in + (threadIdx.x + (10 + j) * 512) * 8);
    float s = 0.f;
    if (MODE == 0) {
        f4 acc[10][4];
        for (int i = 0; i < 10; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 10; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 10; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    } else {
        f16v acc[5][2];
        for (int i = 0; i < 5; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 5; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i * 2 + k], b[j * 2 + k], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 5; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
    const int blocks = 256 * 4, iters = 2000;
    float* out; _Float16* in;
    hipMalloc(&out, blocks * 512 * 4);
    hipMalloc(&in, 512 * 14 * 8 * 2);
    _Float16* h = (_Float16*)malloc(512 * 14 * 8 * 2);
    for (int i = 0; i < 512 * 14 * 8; ++i) h[i] = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
    for (int fill = 0; fill < 2; ++fill) {
        if (fill == 1) for (int i = 0; i < 512 * 14 * 8; ++i) h[i] = (_Float16)0.f;
        hipMemcpy(in, h, 512 * 14 * 8 * 2, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 2; ++mode) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(512), 0, 0, out, in, iters);
                else hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(512), 0, 0, out, in, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flops = (double)blocks * 8 /*waves*/ * iters * 40 * (mode == 0 ? 16384.0 : 16384.0) ;   // 40 x 16x16x32 == 20 x 32x32x16 flops
            printf("%s operands, %s: %.3f ms  %.0f TFLOP/s\n", fill ? "zero" : "random", mode == 0 ? "16x16x32 (10x4 tile)" : "32x32x16 (5x2 tile)", ms, flops / ms / 1e9);
        }
    }
    return 0;
}
