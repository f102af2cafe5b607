// Probe (round 4): the 256 x 320 x 64 GEMM tile's main loop with EIGHT waves of 64 x 160 (the shipped structure, 224 KB of LDS fragment
// reads per k tile) against FOUR waves of 128 x 160 (one wave per SIMD, 320 accumulator registers per lane, 144 KB of fragment reads per
// k tile), the latter with the compiler's MFMA builtin and with inline-asm MFMAs whose accumulators are pinned to AGPRs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/w4_probe.hip -o tools/probes/w4_probe && tools/probes/w4_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#ifndef W4_NO_DMA
#define W4_NO_DMA 0
#endif
#ifndef W4_MJ
#define W4_MJ 8
#define W4_NA 8
#endif
#define W4_STR2(x) #x
#define W4_STR(x) W4_STR2(x)
#include W4_STR(W4_INC)
typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int nx = 8;
    if (nwg < nx) return bid;
    int q = nwg / nx, r = nwg % nx, xcd = bid % nx, idx = bid / nx;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// VARIANT 0: 8 waves (4m x 2n) of 64 x 160, builtin MFMA.  1: 4 waves (2m x 2n) of 128 x 160, builtin.  2: same, asm MFMA with AGPR accumulators.
// 3: as 2 with the operand fragments double-buffered in registers (the next k step's ds_reads fly under this step's 80 MFMAs) and the barrier in mid-tile.
// 4: as 3 with EVERY instruction of the inner blocks in inline asm (w4_body.inc): the source order is the issue order.
template <int VARIANT>
__global__ __launch_bounds__(VARIANT == 0 ? 512 : 256, VARIANT == 0 ? 2 : 1) void tile_kernel(const half_t* __restrict__ X, const half_t* __restrict__ W,
                                                                                               half_t* __restrict__ Y, int M, int N, int K) {
    constexpr int NT = VARIANT == 0 ? 512 : 256;
    constexpr int MJ = VARIANT == 0 ? 4 : W4_MJ;              // 16-row fragments per wave along M
    constexpr int NA = W4_NA;                                 // accumulator columns pinned to AGPRs
    constexpr int NF = 10, BMB = VARIANT == 0 ? 256 : 32 * W4_MJ, BNB = 320, BK = 64, LDSH = 64;
    constexpr int TILE = (BMB + BNB) * LDSH;
    constexpr int RPP = NT / 8;                               // rows staged per pass
    constexpr int XL = BMB / RPP, WL = BNB / RPP;
    __shared__ __attribute__((aligned(16))) half_t smem[2 * TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave & 1, wm = wave >> 1;
    const int l15 = lane & 15, g = lane >> 4;
    const int nt_n = N / BNB;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = lid % nt_n, tm = lid / nt_n;
    const int m0 = tm * BMB, n0 = tn * BNB;
    const int rb = tid >> 3;
    const int kc = (tid & 7) ^ (rb & 7);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto glds16 = [&](const half_t* src, half_t* dst) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    const long xbase = (long)(m0 + rb) * K + kc * 8, wbase = (long)(n0 + rb) * K + kc * 8;
    const long rstride = (long)RPP * K;                       // elements between two staging passes (wave-uniform)
    auto issue = [&](int k0, int buf, int parts) {
        half_t* Xd = smem + buf * TILE + wave_u * 8 * LDSH;
        half_t* Wd = Xd + BMB * LDSH;
        if (parts & 1) {
            const half_t* px = X + xbase + k0;
            if constexpr (VARIANT >= 4) asm volatile("" : "+v"(px));      // opaque per call: the 8 row pointers are NOT hoisted out of the k loop (16 VGPRs)
#pragma unroll
            for (int i = 0; i < XL; ++i) glds16(px + i * rstride, Xd + RPP * i * LDSH);
        }
        if (parts & 2) {
            const half_t* pw = W + wbase + k0;
            if constexpr (VARIANT >= 4) asm volatile("" : "+v"(pw));
#pragma unroll
            for (int i = 0; i < WL; ++i) glds16(pw + i * rstride, Wd + RPP * i * LDSH);
        }
    };
    const int nk = K / BK;
    issue(0, 0, 3);
    // VARIANT 2: 320 accumulator registers exceed the 256 AGPRs: fragments i < NA are pinned to AGPRs, the others to VGPRs
    f4 acc[NF][MJ];
    if constexpr (VARIANT != 4) {
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int j = 0; j < MJ; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (VARIANT == 2 || VARIANT == 3) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < MJ; ++j) asm volatile("" : "+a"(acc[i][j]));
#pragma unroll
        for (int i = NA; i < NF; ++i)
#pragma unroll
            for (int j = 0; j < MJ; ++j) asm volatile("" : "+v"(acc[i][j]));
    }
    const int sw = l15 & 7;
    if constexpr (VARIANT == 4) {
        h8 a0[NF], a1[NF], b[MJ];
        // LDS byte addresses of this lane's fragment rows in buffer 0, per k step (the XOR swizzle makes the two k steps differ by +-64 B)
        const unsigned rowA = (unsigned)((wn * 160 + l15) * LDSH * 2 + BMB * LDSH * 2), rowB = (unsigned)((wm * 16 * MJ + l15) * LDSH * 2);
        const unsigned c0 = (unsigned)(((0 * 4 + g) ^ sw) * 16), c1 = (unsigned)(((1 * 4 + g) ^ sw) * 16);
        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) half_t*)smem;
        const unsigned A0 = lds0 + rowA + c0, A1 = lds0 + rowA + c1, B0 = lds0 + rowB + c0, B1 = lds0 + rowB + c1;
        constexpr unsigned TB = TILE * 2;
        if (nk > 1) issue(BK, 1, 3);
        __syncthreads();
        W4_LOAD0(A0, B0)
        // kt = 0 (peeled: its first MFMAs take C = 0), the steady state without a branch, kt = nk - 1 (peeled: nothing left to prefetch)
#define W4_MID(KT)                                                                                             \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
        __syncthreads();                                                                                       \
        if ((KT) + 2 < nk && !W4_NO_DMA) issue(((KT) + 2) * BK, (KT) & 1, 3);
        {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            W4_BLOCK0_FIRST(A1, B1)
            W4_MID(0)
            const unsigned a0n = A0 + TB, b0n = B0 + TB;
            W4_BLOCK1(a0n, b0n)              // (probe: nk >= 2)
        }
        for (int kt = 1; kt + 1 < nk; ++kt) {
            const unsigned off = (kt & 1) ? TB : 0u, offn = (kt & 1) ? 0u : TB;
            const unsigned a1_ = A1 + off, b1_ = B1 + off, a0n = A0 + offn, b0n = B0 + offn;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            W4_BLOCK0(a1_, b1_)
            W4_MID(kt)
            W4_BLOCK1(a0n, b0n)
        }
        {
            const int kt = nk - 1;
            const unsigned off = (kt & 1) ? TB : 0u;
            const unsigned a1_ = A1 + off, b1_ = B1 + off;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            W4_BLOCK0(a1_, b1_)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            W4_BLOCK1_LAST
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");       // hipcc does not know the statements above are MFMAs: no wait states of its own before it reads D
    } else if constexpr (VARIANT == 3) {
        auto mfmas = [&](const h8 (&a)[NF], const h8 (&b)[MJ]) {
#pragma unroll
            for (int j = 0; j < MJ; ++j) {
#pragma unroll
                for (int i = 0; i < NA; ++i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(a[i]), "v"(b[j]));
#pragma unroll
                for (int i = NA; i < NF; ++i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(a[i]), "v"(b[j]));
            }
        };
        auto frags = [&](int buf, int ks, h8 (&a)[NF], h8 (&b)[MJ]) {
            const half_t* Xs = smem + buf * TILE;
            const half_t* Ws = Xs + BMB * LDSH;
            const int ch = ((ks * 4 + g) ^ sw) * 8;
#pragma unroll
            for (int i = 0; i < NF; ++i) a[i] = *reinterpret_cast<const h8*>(&Ws[(wn * 160 + i * 16 + l15) * LDSH + ch]);
#pragma unroll
            for (int j = 0; j < MJ; ++j) b[j] = *reinterpret_cast<const h8*>(&Xs[(wm * 16 * MJ + j * 16 + l15) * LDSH + ch]);
        };
        h8 a0[NF], b0[MJ], a1[NF], b1[MJ];
        if (nk > 1) issue(BK, 1, 3);
        __syncthreads();                      // (vmcnt(0): both tiles landed — the steady state below only waits for one)
        frags(0, 0, a0, b0);
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            frags(cur, 1, a1, b1);
            mfmas(a0, b0);
            __syncthreads();                  // tile kt+1 landed everywhere; every wave holds its last fragments of tile kt: buffer `cur` is free
            if (kt + 2 < nk) issue((kt + 2) * BK, cur, 3);
            if (kt + 1 < nk) frags(cur ^ 1, 0, a0, b0);
            mfmas(a1, b1);
        }
    } else
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
        const bool more = kt + 1 < nk;
        if (more) issue((kt + 1) * BK, (kt + 1) & 1, 1);
        const half_t* Xs = smem + (kt & 1) * TILE;
        const half_t* Ws = Xs + BMB * LDSH;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks == 1 && more) issue((kt + 1) * BK, (kt + 1) & 1, 2);
            const int ch = ((ks * 4 + g) ^ sw) * 8;
            h8 a[NF];
#pragma unroll
            for (int i = 0; i < NF; ++i) a[i] = *reinterpret_cast<const h8*>(&Ws[(wn * 160 + i * 16 + l15) * LDSH + ch]);
#pragma unroll
            for (int j = 0; j < MJ; ++j) {
                const h8 b = *reinterpret_cast<const h8*>(&Xs[(wm * 16 * MJ + j * 16 + l15) * LDSH + ch]);
                if constexpr (VARIANT >= 2) {
#pragma unroll
                    for (int i = 0; i < NA; ++i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(a[i]), "v"(b));
#pragma unroll
                    for (int i = NA; i < NF; ++i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(a[i]), "v"(b));
                } else {
#pragma unroll
                    for (int i = 0; i < NF; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b, acc[i][j], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
        const int m = m0 + wm * 16 * MJ + j * 16 + l15;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const f4 v = acc[i][j];
            h4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            *reinterpret_cast<h4*>(Y + (long)m * N + n0 + wn * 160 + i * 16 + g * 4) = o;
        }
    }
}

template <int V>
static float run(const half_t* X, const half_t* W, half_t* Y, int M, int N, int K, int it) {
    dim3 grid((M / (V == 0 ? 256 : 32 * W4_MJ)) * (N / 320)), block(V == 0 ? 512 : 256);
    hipLaunchKernelGGL((tile_kernel<V>), grid, block, 0, 0, X, W, Y, M, N, K);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL((tile_kernel<V>), grid, block, 0, 0, X, W, Y, M, N, K);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / it;
}

int main() {
    const int shapes[][3] = {{49152, 1280, 1280}, {49152, 640, 2560}, {196608, 320, 1280}, {49152, 640, 5760}, {196608, 960, 320}, {12288, 1280, 5120}};
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        std::vector<half_t> hx((size_t)M * K), hw((size_t)N * K);
        srand(1);
        for (auto& v : hx) v = (half_t)((rand() % 2001 - 1000) / 1000.f);
        for (auto& v : hw) v = (half_t)((rand() % 2001 - 1000) / 1000.f * 0.05f);
        half_t *X, *W, *Y0, *Y1;
        CK(hipMalloc(&X, hx.size() * 2)); CK(hipMalloc(&W, hw.size() * 2)); CK(hipMalloc(&Y0, (size_t)M * N * 2)); CK(hipMalloc(&Y1, (size_t)M * N * 2));
        CK(hipMemcpy(X, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
        const double fl = 2.0 * M * N * K;
        float t0 = run<0>(X, W, Y0, M, N, K, 10);
        float t1 = run<1>(X, W, Y1, M, N, K, 10);
        std::vector<half_t> y0((size_t)M * N), y1((size_t)M * N);
        CK(hipMemcpy(y0.data(), Y0, y0.size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(y1.data(), Y1, y1.size() * 2, hipMemcpyDeviceToHost));
        size_t bad1 = 0;
        for (size_t i = 0; i < y0.size(); ++i) bad1 += (y0[i] != y1[i]);
        float t2 = run<2>(X, W, Y1, M, N, K, 10);
        CK(hipMemcpy(y1.data(), Y1, y1.size() * 2, hipMemcpyDeviceToHost));
        size_t bad2 = 0;
        for (size_t i = 0; i < y0.size(); ++i) bad2 += (y0[i] != y1[i]);
        float t4 = run<4>(X, W, Y1, M, N, K, 10);
        CK(hipMemcpy(y1.data(), Y1, y1.size() * 2, hipMemcpyDeviceToHost));
        size_t bad4 = 0;
        for (size_t i = 0; i < y0.size(); ++i) bad4 += (y0[i] != y1[i]);
        float t3 = run<3>(X, W, Y1, M, N, K, 10);
        CK(hipMemcpy(y1.data(), Y1, y1.size() * 2, hipMemcpyDeviceToHost));
        size_t bad3 = 0;
        for (size_t i = 0; i < y0.size(); ++i) bad3 += (y0[i] != y1[i]);
        float t0b = run<0>(X, W, Y0, M, N, K, 10);
        printf("M=%6d N=%5d K=%5d: 8 waves %.3f / %.3f ms (%.0f TF) | 4 waves builtin %.3f ms (%.0f TF, %zu diffs) | 4 waves asm+AGPR %.3f ms (%.0f TF, %zu diffs) | + register double buffer %.3f ms (%.0f TF, %zu diffs) | all-asm schedule %.3f ms (%.0f TF, %zu diffs)\n", M, N, K,
               t0, t0b, fl / t0 / 1e9, t1, fl / t1 / 1e9, bad1, t2, fl / t2 / 1e9, bad2, t3, fl / t3 / 1e9, bad3, t4, fl / t4 / 1e9, bad4);
        CK(hipFree(X)); CK(hipFree(W)); CK(hipFree(Y0)); CK(hipFree(Y1));
    }
    return 0;
}
