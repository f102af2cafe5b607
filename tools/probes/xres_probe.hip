// Micro-probe (not part of the product): an X-RESIDENT GEGLU projection for K = 320 (the 64x64-level FF1: M = 196 608, N = 2 560).
//
// Today's 256x320 tile spends 13 of its 23 us at K = 320 outside the k loop (prologue / drain, GELU, store tail), and nothing of it
// overlaps: every wave of the block is in its epilogue at once, and VALU work of one wave does not overlap MFMAs of another wave of the
// same SIMD (coissue_probe).  What does overlap is VALU work issued BETWEEN the MFMAs of the same wave.  Here a block owns 128 rows, keeps
// their 320 activations in LDS (80 KB, loaded once) and walks the ten 256-column tiles of N with only the weights streaming (two 32 KB
// buffers); waves 2(m) x 4(n) with 64x64 register tiles = 64 accumulators, so a second set fits and the GELU + stores of column tile t
// sit in the k loop of column tile t+1.  Weight rows are interleaved [x0 x1 g0 g1 ...] so that a lane holds whole (x, gate) pairs.
//
//   xres_probe            correctness (sampled against a plain fp32 kernel) + time, with and without the deferred epilogue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#pragma clang diagnostic ignored "-Wunused-result"
typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __attribute__((aligned(256))) half_t zero_page[128];

__device__ __forceinline__ void geglu2x2(f2 xa, f2 ga, f2 xb, f2 gb, f2& ya, f2& yb) {
    f2 ca, cb;
    ca.x = __builtin_amdgcn_fmed3f(ga.x, -5.f, 5.f); ca.y = __builtin_amdgcn_fmed3f(ga.y, -5.f, 5.f);
    cb.x = __builtin_amdgcn_fmed3f(gb.x, -5.f, 5.f); cb.y = __builtin_amdgcn_fmed3f(gb.y, -5.f, 5.f);
    const f2 sa = __builtin_elementwise_fma(ca * 0.08f, ca, f2{-1.f, -1.f});
    const f2 sb = __builtin_elementwise_fma(cb * 0.08f, cb, f2{-1.f, -1.f});
    constexpr float c[13] = {2.827276369e-01f, -1.405918177e-01f, 1.030358595e-01f, -8.090256480e-02f, 6.295351729e-02f, -4.642625655e-02f,
                             3.247217962e-02f, -2.261424982e-02f, 1.353305501e-02f, -5.053833458e-03f, 2.749192302e-03f, -3.353461957e-03f,
                             1.470752778e-03f};
    f2 pa = f2{c[12], c[12]}, pb = f2{c[12], c[12]};
#pragma unroll
    for (int i = 11; i >= 0; --i) {
        pa = __builtin_elementwise_fma(pa, sa, f2{c[i], c[i]});
        pb = __builtin_elementwise_fma(pb, sb, f2{c[i], c[i]});
    }
    const f2 Sa = ca * pa, Sb = cb * pb;
    const f2 ta = (xa * 0.5f) * ga, tb = (xb * 0.5f) * gb;
    ya = __builtin_elementwise_fma(ta, Sa, ta);
    yb = __builtin_elementwise_fma(tb, Sb, tb);
}

constexpr int BM = 128, BN = 256, BK = 64, KDIM = 320, NKT = KDIM / BK;
constexpr int XS = NKT * BM * BK;            // halfs: 80 KB
constexpr int WT = BN * BK;                  // halfs per weight buffer: 32 KB
constexpr int SLAB = 16 * 40;                // halfs per wave: 16 rows x 32 outputs (+8 pad)

struct P {
    const half_t* X; const half_t* Wp; const half_t* bias; half_t* Y;
    int M, N;      // N = 2 * hidden, rows of Wp in the interleaved order
};

// DEFER = 1: epilogue of column tile t inside the k loop of tile t+1;  0: right after its own k loop (same code, for comparison)
template <int DEFER>
__global__ __launch_bounds__(512, 2) void xres_kernel(P p) {
    __shared__ __attribute__((aligned(16))) half_t smem[XS + 2 * WT + 8 * SLAB];
    half_t* const Xs = smem;
    half_t* const Wb = smem + XS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int l15 = lane & 15, g = lane >> 4;
    half_t* const slab = smem + XS + 2 * WT + wave * SLAB;
    const int m0 = blockIdx.x * BM;
    const int NT = p.N / BN;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto glds16 = [&](const half_t* src, half_t* dst) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    const int rb = tid >> 3, kc = (tid & 7) ^ (rb & 7);
    // ---- the block's activations, once: k tile kt = rows of 128 B, chunk-swizzled
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + rb + 64 * i;
            glds16(m < p.M ? p.X + (long)m * KDIM + kt * BK + kc * 8 : zero_page, Xs + kt * BM * BK + (64 * i + wave_u * 8) * BK);
        }
    auto issue_w = [&](int nt, int kt, int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = nt * BN + rb + 64 * i;
            glds16(p.Wp + (long)n * KDIM + kt * BK + kc * 8, Wb + buf * WT + (64 * i + wave_u * 8) * BK);
        }
    };
    issue_w(0, 0, 0);
    const int sw = l15 & 7;
    f4 accA[4][4], accB[4][4];

    // epilogue unit j of a finished column tile: rows wm*64 + j*16 + l15, hidden columns nt*128 + wn*32 + i*8 + g*2 + {0, 1}
    // bias of a column tile's 16 (x, gate) pairs of this lane, fetched at the START of the k loop that carries the tile's epilogue
    auto load_bias = [&](int nt, h2 (&bb)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int hcol = nt * 128 + wn * 32 + i * 8 + g * 2;
            bb[2 * i] = p.bias ? *reinterpret_cast<const h2*>(p.bias + hcol) : h2{0, 0};
            bb[2 * i + 1] = p.bias ? *reinterpret_cast<const h2*>(p.bias + p.N / 2 + hcol) : h2{0, 0};
        }
    };
    // half an epilogue unit: fragments i0, i0 + 1 of row fragment j -> GELU -> slab; the second half also moves the 16 x 32 piece out
    auto epi_half = [&](const f4 (&acc)[4][4], const h2 (&bb)[8], int nt, int j, int i) {
        const float bx0 = (float)bb[2 * i][0], bx1 = (float)bb[2 * i][1], bg0 = (float)bb[2 * i + 1][0], bg1 = (float)bb[2 * i + 1][1];
        const float cx0 = (float)bb[2 * i + 2][0], cx1 = (float)bb[2 * i + 2][1], cg0 = (float)bb[2 * i + 3][0], cg1 = (float)bb[2 * i + 3][1];
        f2 ya, yb;
        geglu2x2(f2{acc[i][j][0] + bx0, acc[i][j][1] + bx1}, f2{acc[i][j][2] + bg0, acc[i][j][3] + bg1},
                 f2{acc[i + 1][j][0] + cx0, acc[i + 1][j][1] + cx1}, f2{acc[i + 1][j][2] + cg0, acc[i + 1][j][3] + cg1}, ya, yb);
        *reinterpret_cast<h2*>(&slab[l15 * 40 + i * 8 + g * 2]) = h2{(half_t)ya.x, (half_t)ya.y};
        *reinterpret_cast<h2*>(&slab[l15 * 40 + (i + 1) * 8 + g * 2]) = h2{(half_t)yb.x, (half_t)yb.y};
        if (i == 2) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int row = lane >> 2, c = lane & 3;
            const h8 v = *reinterpret_cast<const h8*>(&slab[row * 40 + c * 8]);
            const int m = m0 + wm * 64 + j * 16 + row;
            if (m < p.M) *reinterpret_cast<h8*>(p.Y + (long)m * (p.N / 2) + nt * 128 + wn * 32 + c * 8) = v;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    };
    auto epi_unit = [&](const f4 (&acc)[4][4], const h2 (&bb)[8], int nt, int j) {
        epi_half(acc, bb, nt, j, 0);
        epi_half(acc, bb, nt, j, 2);
    };

    // one column tile: 5 k tiles into `cur`; DEFER: unit kt of `held` (column tile nt - 1) rides in k tile kt
    auto tile = [&](f4 (&cur)[4][4], const f4 (&held)[4][4], int nt, bool have_held) {
        h2 bb[8];
        load_bias(DEFER ? (nt > 0 ? nt - 1 : 0) : nt, bb);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) cur[i][j] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            const int step = nt * NKT + kt;
            __syncthreads();                  // vmcnt(0) + barrier: weight tile `step` (and, at step 0, the activations) landed; the other buffer is free
            if (kt + 1 < NKT) issue_w(nt, kt + 1, (step + 1) & 1);
            else if (nt + 1 < NT) issue_w(nt + 1, 0, (step + 1) & 1);
            const half_t* Ws = Wb + (step & 1) * WT;
            const half_t* Xk = Xs + kt * BM * BK;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int ch = ((ks * 4 + g) ^ sw) * 8;
                h8 a[4], b[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const h8*>(&Ws[(wn * 64 + i * 16 + l15) * BK + ch]);
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const h8*>(&Xk[(wm * 64 + j * 16 + l15) * BK + ch]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) cur[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], cur[i][j], 0, 0, 0);
                if (DEFER && have_held && kt < 4) {
                    epi_half(held, bb, nt - 1, kt, ks * 2);
                    if (DEFER == 2) {
#pragma unroll
                        for (int q = 0; q < 16; ++q) {        // one MFMA, then three VALU: the half unit rides between this k step's MFMAs
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                        }
                    }
                }
            }
        }
        if (!DEFER) {
#pragma unroll
            for (int j = 0; j < 4; ++j) epi_unit(cur, bb, nt, j);
        }
    };
    for (int nt = 0; nt < NT; nt += 2) {
        tile(accA, accB, nt, nt > 0);
        if (nt + 1 < NT) tile(accB, accA, nt + 1, true);
    }
    if (DEFER) {
        h2 bb[8];
        load_bias(NT - 1, bb);
        if (NT & 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) epi_unit(accA, bb, NT - 1, j);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) epi_unit(accB, bb, NT - 1, j);
        }
    }
}

// plain reference for sampled outputs: fp32 accumulation of the fp16 operands, exact erf GELU
__global__ void ref_kernel(const half_t* X, const half_t* W, const half_t* bias, const int* samp, float* out, int ns, int N) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= ns) return;
    const int m = samp[2 * s], h = samp[2 * s + 1];
    float ax = bias ? (float)bias[h] : 0.f, ag = bias ? (float)bias[N / 2 + h] : 0.f;
    for (int k = 0; k < KDIM; ++k) {
        ax += (float)X[(long)m * KDIM + k] * (float)W[(long)h * KDIM + k];
        ag += (float)X[(long)m * KDIM + k] * (float)W[(long)(N / 2 + h) * KDIM + k];
    }
    out[s] = ax * 0.5f * ag * (1.f + erff(ag * 0.70710678f));
}

int main() {
    const int M = 196608, N = 2560, H = N / 2;
    std::vector<half_t> hx((size_t)M * KDIM), hw((size_t)N * KDIM), hwp((size_t)N * KDIM), hb(N);
    srand(1);
    for (auto& v : hx) v = (half_t)((rand() % 2001 - 1000) / 1000.0f);
    for (auto& v : hw) v = (half_t)((rand() % 2001 - 1000) / 1000.0f * 0.056f);
    for (auto& v : hb) v = (half_t)((rand() % 2001 - 1000) / 5000.0f);
    // interleave: row n' of Wp (column tile nt, wave column wn, fragment i, lane group g, r): r < 2 -> x row h, else gate row H + h
    for (int n = 0; n < N; ++n) {
        const int nt = n / 256, wn = (n % 256) / 64, i = (n % 64) / 16, gg = (n % 16) / 4, r = n % 4;
        const int h = nt * 128 + wn * 32 + i * 8 + gg * 2 + (r & 1);
        const int src = r < 2 ? h : H + h;
        for (int k = 0; k < KDIM; ++k) hwp[(size_t)n * KDIM + k] = hw[(size_t)src * KDIM + k];
    }
    half_t *dx, *dw, *dwp, *db, *dy;
    hipMalloc(&dx, hx.size() * 2); hipMalloc(&dw, hw.size() * 2); hipMalloc(&dwp, hw.size() * 2); hipMalloc(&db, N * 2); hipMalloc(&dy, (size_t)M * H * 2);
    hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dwp, hwp.data(), hw.size() * 2, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), N * 2, hipMemcpyHostToDevice);
    const int ns = 8192;
    std::vector<int> hs(2 * ns);
    for (int s = 0; s < ns; ++s) { hs[2 * s] = rand() % M; hs[2 * s + 1] = rand() % H; }
    int* ds; float* dref;
    hipMalloc(&ds, hs.size() * 4); hipMalloc(&dref, ns * 4);
    hipMemcpy(ds, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(ref_kernel, dim3((ns + 255) / 256), dim3(256), 0, 0, dx, dw, db, ds, dref, ns, N);
    std::vector<float> href(ns);
    hipMemcpy(href.data(), dref, ns * 4, hipMemcpyDeviceToHost);
    P p{dx, dwp, db, dy, M, N};
    for (int defer = 0; defer <= 2; ++defer) {
        hipMemset(dy, 0, (size_t)M * H * 2);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e30f;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0);
            if (defer == 2) hipLaunchKernelGGL(xres_kernel<2>, dim3(M / BM), dim3(512), 0, 0, p);
            else if (defer) hipLaunchKernelGGL(xres_kernel<1>, dim3(M / BM), dim3(512), 0, 0, p);
            else hipLaunchKernelGGL(xres_kernel<0>, dim3(M / BM), dim3(512), 0, 0, p);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        hipError_t err = hipGetLastError();
        std::vector<half_t> hy(1);
        double maxerr = 0, maxref = 0;
        for (int s = 0; s < ns; ++s) {
            half_t v; hipMemcpy(&v, dy + (size_t)hs[2 * s] * H + hs[2 * s + 1], 2, hipMemcpyDeviceToHost);
            maxerr = fmax(maxerr, fabs((double)(float)v - href[s])); maxref = fmax(maxref, fabs((double)href[s]));
        }
        printf("deferred epilogue %d: %.3f ms  %.0f TFLOP/s  (shipped 256x320 tile: 0.556 ms = 580)   max |err| %.2e of max |ref| %.2f  (%s)\n", defer, best,
               2.0 * M * N * KDIM / best / 1e9, maxerr, maxref, hipGetErrorString(err));
    }
    return 0;
}
