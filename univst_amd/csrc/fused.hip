// Fused row-local operators of a transformer block (round 5).
//
// attn2_fused_kernel: the whole text cross-attention of a block in ONE launch
//     Y = to_out( softmax( (LN(X) Wq^T) K^T ) V ) + bias + X          (attention.py:321-327: norm2 -> attn2 -> residual)
// replacing three launches (q projection, 77-key attention, out projection + residual) and the 2 x |Q| + 2 x |O| + |X| bytes of HBM
// traffic between them.  A block owns 64 rows and keeps ONE 64 x C fp16 tile in LDS that is, in turn, the input rows (LDS-DMA, XOR-
// swizzled k tiles), the queries, the attention outputs and the output tile:
//   phase A   Q = LN(X) Wq'^T      wave w computes columns [w*C/4, (w+1)*C/4) = two whole heads for all 64 rows; the weight fragments
//                                   come straight from memory / L2 in MFMA operand order (`univst_frag_pack`: one contiguous 1 KB load per
//                                   fragment — a weight row slice is used by exactly one wave of the block, so staging it in LDS would
//                                   only add a round trip), three k steps in flight; LayerNorm folded into the epilogue (gemm.hip)
//   phase B   per head of the wave: S^T = K_h Q_h^T with K_h (5 key fragments) and V_h^T held in registers as MFMA A operands
//             (`kv_frag_pack_kernel` lays the text K | V of every (branch, head) out in operand order once per call), all 80 scores of a
//             row in registers (exact max, no online rescale), P^T packed in place as the B operand of O^T = V^T P^T — the body of
//             attn_text_kernel (attention.hip) with Q read from and O written to the wave's own columns of the tile: no block barrier
//   phase C   Y = O Wo^T + bias    as phase A; the tile is then rewritten with Y and leaves row-contiguously (16 B per lane) with the
//             residual rows added (re-read: L2 / Infinity Cache hits) and the row statistics of the stored values for a following
//             folded LayerNorm (GemmParams::stats_out layout)
// Arithmetic: fp32 accumulation; Q, P, O and Y are rounded to fp16 where the unfused graph rounds them, plus Y once before the residual
// add (the reference's own order: Linear output in fp16, then `+ hidden_states`).
#include "common.h"
#include <stdlib.h>

#include "kernels.h"

namespace {

// phase timestamps of every wave (tools/probes/attn2_probe.hip only: -DUV_A2_TRACE adds Attn2Params::trace)
#ifdef UV_A2_TRACE
#define A2T(i)                                                                                                       \
    do {                                                                                                             \
        if (p.trace && (threadIdx.x & 63) == 0) p.trace[((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (i)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define A2T(i)
#endif

__device__ __attribute__((aligned(256))) half_t fz_zero_page[128];

__device__ __forceinline__ float max3f_(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
typedef __fp16 fh2_ __attribute__((ext_vector_type(2)));

// halfs offset of (row, col) in the 64-row tile: k-tile major [C/64][64 rows][64 halfs], 16-byte chunk index XOR-swizzled with row & 7
__device__ __forceinline__ int tile_off(int row, int col) {
    return (col >> 6) * 4096 + row * 64 + ((((col & 63) >> 3) ^ (row & 7)) << 3) + (col & 7);
}

// W [N][K] -> MFMA A-operand order [N/16][K/32][64 lanes][8]: lane (l15, g) holds W[nf*16 + l15][ks*32 + g*8 .. +8]
__global__ void frag_pack_kernel(const half_t* __restrict__ W, half_t* __restrict__ out, int N, int K) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one 16-byte piece
    const int KS = K / 32;
    if (i >= (long)(N / 16) * KS * 64) return;
    const int lane = (int)(i & 63);
    const long f = i >> 6;
    const int ks = (int)(f % KS), nf = (int)(f / KS);
    *reinterpret_cast<h8*>(out + i * 8) = *reinterpret_cast<const h8*>(W + (long)(nf * 16 + (lane & 15)) * K + ks * 32 + (lane >> 4) * 8);
}

// text K | V rows [B*T][2C] -> per (branch, head): K fragments [5][KS][64][8] (lane (l15, g): key kf*16 + l15, d = ks*32 + g*8 ..) then V^T
// fragments [DV16][3][64][8] (lane (l15, g): d = dv*16 + l15; elements 0..3 = keys t*32 + g*4 + e, 4..7 = keys t*32 + 16 + g*4 + e - 4 —
// the order in which a lane's score registers sit); zero beyond T keys / D columns.
template <int D>
__global__ __launch_bounds__(256) void kv_frag_pack_kernel(const half_t* __restrict__ kv, half_t* __restrict__ out, int T, int C, int heads) {
    constexpr int KS = (D + 31) / 32, DV16 = (D + 15) / 16, NK = 5 * KS * 64, NV = DV16 * 3 * 64;
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    half_t* o = out + (long)blockIdx.x * (NK + NV) * 8;
    for (int i = threadIdx.x; i < NK + NV; i += 256) {
        h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        const int lane = i & 63, l15 = lane & 15, g = lane >> 4;
        if (i < NK) {
            const int f = i >> 6, ks = f % KS, kf = f / KS;
            const int key = kf * 16 + l15, d = ks * 32 + g * 8;
            if (key < T && d < D) v = *reinterpret_cast<const h8*>(kv + (long)(b * T + key) * 2 * C + h * D + d);
        } else {
            const int f = (i - NK) >> 6, t = f % 3, dv = f / 3;
            const int d = dv * 16 + l15;
            if (d < D) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int key = t * 32 + (e >> 2) * 16 + g * 4 + (e & 3);
                    if (key < T) v[e] = kv[(long)(b * T + key) * 2 * C + C + h * D + d];
                }
            }
        }
        *reinterpret_cast<h8*>(o + (long)i * 8) = v;
    }
}

// OCC: resident blocks per CU the register budget is set for (2: 256 VGPRs, weight ring three k steps deep; 3: 168 VGPRs, ring two deep).  The tile is
// the only LDS object (40 KB at C = 320: the row-statistics scratch reuses it once every thread holds its output chunks), so LDS allows either.
// PRE (round 5, second step): the SELF-attention's out projection in front — X holds the attention output rows, phase P computes
//     H2 = X Wp^T + bias_p + Rp          (attention.py:316-319: attn1.to_out + hidden_states)
// into the tile AND into a second tile that stays as the residual of the last phase; the LayerNorm statistics of H2 are taken from the tile by the
// waves themselves (no producer statistics, no block-wide scratch: lane (l15, g) sums row 16 g + l15 and the four rows a lane needs come back by
// shuffle), so H2 — which nothing else reads — never reaches HBM and the out-projection launch disappears.  LDS: two 40 KB tiles = 80 KB, two blocks per CU
// still fit (tools/probes: 0.182 ms with either footprint).
template <int C, int D, bool LNF, int OCC = 2, bool PRE = false>
__global__ __launch_bounds__(256, OCC) void attn2_fused_kernel(Attn2Params p) {
    static_assert(!PRE || LNF, "the fused out projection computes the LayerNorm statistics itself");
    constexpr int BM = 64, NKT = C / 64, KSTEPS = C / 32, NFW = C / 64;       // NFW: 16-column fragments per wave (C/4 columns)
    constexpr int KS = (D + 31) / 32, DV16 = (D + 15) / 16, NKF = 5;
    constexpr int KVF = (NKF * KS + DV16 * 3) * 64 * 8;                       // halfs per (branch, head) in kvf
    constexpr int CH = C / 8;                                                 // 16-byte chunks per row
    constexpr int TILE = BM * C;
    static_assert((C / 4) % D == 0 && (C / 4) / D == 2, "a wave owns two whole heads");
    constexpr int RING = OCC >= 3 ? 2 : 3;                                    // weight fragments in flight: RING k steps
#ifdef UV_A2_LDS2            // probe only: does a second 40 KB tile still leave two blocks per CU (2 x 80 KB = all of the LDS)?  (it does)
    __shared__ __attribute__((aligned(16))) half_t smem[2 * TILE];
#else
    __shared__ __attribute__((aligned(16))) half_t smem[PRE ? 2 * TILE : TILE];
#endif
    half_t* const Rt = smem + TILE;                                           // PRE: H2, kept for the residual add of the last phase
    half_t* const T = smem;
    float2* const scr = reinterpret_cast<float2*>(smem);                      // row-statistics scratch: the tile, after the last read of it (BM * CH float2 = half the tile)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int m0 = blockIdx.x * BM;
    const h8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const f4 z4 = {0.f, 0.f, 0.f, 0.f};

    A2T(0);
    // ---- phase 0: the block's rows -> LDS, once (k tile kt: 64 rows of 128 B, chunks XOR-swizzled with row & 7)
    {
        const int rb = tid >> 3;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = i * 32 + rb, kc = (tid & 7) ^ (row & 7);
                const half_t* src = (m0 + row < p.M) ? p.X + (long)(m0 + row) * p.ldx + kt * 64 + kc * 8 : fz_zero_page;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(T + kt * 4096 + (i * 32 + wave_u * 8) * 64), 16, 0, 0);
            }
    }
    // ---- projection of the tile by a fragment-packed weight: acc[i][j] = columns wave*C/4 + i*16 + g*4 + r of rows j*16 + l15
    f4 acc[NFW][4];
    auto project = [&](const half_t* Wf, bool wait_tile, int tslot) {
        const h8* wp = reinterpret_cast<const h8*>(Wf) + (long)(wave * NFW) * KSTEPS * 64 + lane;
        h8 a[RING][NFW];
#pragma unroll
        for (int s = 0; s < RING - 1; ++s)
#pragma unroll
            for (int i = 0; i < NFW; ++i) a[s][i] = wp[(long)(i * KSTEPS + s) * 64];
#pragma unroll
        for (int i = 0; i < NFW; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = z4;
        if (wait_tile) __syncthreads();              // vmcnt(0) + barrier: the rows landed (phase A) / every wave's O columns are written (phase C)
        A2T(tslot);
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            if (ks + RING - 1 < KSTEPS) {
#pragma unroll
                for (int i = 0; i < NFW; ++i) a[(ks + RING - 1) % RING][i] = wp[(long)(i * KSTEPS + ks + RING - 1) * 64];
            }
            h8 b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const h8*>(&T[tile_off(j * 16 + l15, ks * 32 + g * 8)]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < NFW; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ks % RING][i], b[j], acc[i][j], 0, 0, 0);
        }
    };

    // ---- phase A: Q = LN(X) Wq'^T
    float2 lnrow[LNF ? 4 : 1];
    if constexpr (PRE) {
        // ---- phase P: H2 = X Wp^T + bias_p + Rp.  The residual rows are requested before the k loop (MFMA layout: 8 bytes per fragment and lane)
        const int nb = wave * (C / 4) + g * 4;
        h4 hres[NFW][4];
#pragma unroll
        for (int i = 0; i < NFW; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = m0 + j * 16 + l15;
                hres[i][j] = m < p.M ? *reinterpret_cast<const h4*>(p.Rp + (long)m * p.ldrp + nb + i * 16) : h4{0, 0, 0, 0};
            }
        project(p.Wp_f, true, 1);
        __syncthreads();                             // every wave is done reading the attention output: the tile becomes H2
#pragma unroll
        for (int i = 0; i < NFW; ++i) {
            h4 bv = {0, 0, 0, 0};
            if (p.bias_p) bv = *reinterpret_cast<const h4*>(p.bias_p + nb + i * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                h4 hv;
#pragma unroll
                for (int r = 0; r < 4; ++r) hv[r] = (half_t)(acc[i][j][r] + (float)bv[r] + (float)hres[i][j][r]);      // one rounding, as the unfused epilogue
                const int off = tile_off(j * 16 + l15, nb + i * 16);
                *reinterpret_cast<h4*>(&T[off]) = hv;
                *reinterpret_cast<h4*>(&Rt[off]) = hv;
            }
        }
        __syncthreads();                             // H2 complete
        // LayerNorm statistics of the stored fp16 rows: lane (l15, g) takes row 16 g + l15 (every wave all 64 rows: no cross-wave exchange)
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c8 = 0; c8 < CH; ++c8) {
            const h8 v = *reinterpret_cast<const h8*>(&T[tile_off(g * 16 + l15, c8 * 8)]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = (float)v[e];
                s1 += t;
                s2 = fmaf(t, t, s2);
            }
        }
        const float mean = s1 * (1.f / C);
        const float rstd = rsqrtf(fmaxf(fmaf(-mean, mean, s2 * (1.f / C)), 0.f) + p.ln_eps);
#pragma unroll
        for (int j = 0; j < 4; ++j) lnrow[j] = float2{__shfl(mean, j * 16 + l15, 64), __shfl(rstd, j * 16 + l15, 64)};
    } else if constexpr (LNF) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + j * 16 + l15;
            float s1 = 0.f, s2 = 0.f;
            if (m < p.M) {
                const float2* sp = reinterpret_cast<const float2*>(p.ln_stats) + (long)m * p.ln_slots;
                for (int e = 0; e < p.ln_slots; ++e) {
                    const float2 t = sp[e];
                    s1 += t.x;
                    s2 += t.y;
                }
            }
            const float mean = s1 * (1.f / C);
            const float var = fmaxf(fmaf(-mean, mean, s2 * (1.f / C)), 0.f);
            lnrow[j] = float2{mean, rsqrtf(var + p.ln_eps)};
        }
    }
    project(p.Wq_f, true, 1);
    A2T(2);
    __syncthreads();                                 // every wave is done reading X: the tile becomes Q
    {
        const int nb = wave * (C / 4) + g * 4;
#pragma unroll
        for (int i = 0; i < NFW; ++i) {
            f4 ws = z4, cb = z4;
            if constexpr (LNF) {
                ws = *reinterpret_cast<const f4*>(p.ln_wsum + nb + i * 16);
                cb = *reinterpret_cast<const f4*>(p.ln_bias + nb + i * 16);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                h4 q;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[i][j][r];
                    if constexpr (LNF) v = fmaf(lnrow[j].y, fmaf(-lnrow[j].x, ws[r], v), cb[r]);
                    q[r] = (half_t)v;
                }
                *reinterpret_cast<h4*>(&T[tile_off(j * 16 + l15, nb + i * 16)]) = q;
            }
        }
    }
    A2T(3);
    // ---- phase B: the wave's two heads, 32 query rows at a time (wave-private columns of the tile: no block barrier)
    {
        const int br = m0 / p.rows_per_branch;
        const float c = p.q_prescaled ? 1.f : p.scale_log2e;
#pragma unroll 1
        for (int hh = 0; hh < 2; ++hh) {
            const int h = wave * 2 + hh;
            const h8* kvp = reinterpret_cast<const h8*>(p.kvf + (long)(br * p.heads + h) * KVF) + lane;
            h8 kf[NKF][KS], vf[DV16][3];
#pragma unroll
            for (int kfi = 0; kfi < NKF; ++kfi)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) kf[kfi][ks] = kvp[(kfi * KS + ks) * 64];
#pragma unroll
            for (int dv = 0; dv < DV16; ++dv)
#pragma unroll
                for (int t = 0; t < 3; ++t) vf[dv][t] = kvp[(NKF * KS + dv * 3 + t) * 64];
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
                h8 qf[2][KS];
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const int dc = ks * 32 + g * 8;
                        qf[qb][ks] = dc < D ? *reinterpret_cast<const h8*>(&T[tile_off(half * 32 + qb * 16 + l15, h * D + dc)]) : zero8;
                    }
                f4 sc[NKF][2];
#pragma unroll
                for (int kfi = 0; kfi < NKF; ++kfi)
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb) {
                        f4 a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kfi][0], qf[qb][0], z4, 0, 0, 0);
#pragma unroll
                        for (int ks = 1; ks < KS; ++ks) a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kfi][ks], qf[qb][ks], a, 0, 0, 0);
                        sc[kfi][qb] = a;
                    }
#pragma unroll
                for (int kfi = 0; kfi < NKF; ++kfi)
                    if (kfi * 16 + 16 > p.Nkv) {
#pragma unroll
                        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (kfi * 16 + g * 4 + r >= p.Nkv) sc[kfi][qb][r] = -INFINITY;
                    }
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    float m = max3f_(sc[0][qb][0], sc[0][qb][1], sc[0][qb][2]);
                    m = fmaxf(m, sc[0][qb][3]);
#pragma unroll
                    for (int kfi = 1; kfi < NKF; ++kfi) {
                        m = max3f_(m, sc[kfi][qb][0], sc[kfi][qb][1]);
                        m = max3f_(m, sc[kfi][qb][2], sc[kfi][qb][3]);
                    }
                    m = fmaxf(m, __shfl_xor(m, 16, 64));
                    m = fmaxf(m, __shfl_xor(m, 32, 64));
                    const float mc = -m * c;
                    float l = 0.f;
                    union { fh2_ h2v[4]; h8 v; } u[3];
                    u[2].v = zero8;
#pragma unroll
                    for (int kfi = 0; kfi < NKF; ++kfi) {
                        const float e0 = __builtin_amdgcn_exp2f(fmaf(sc[kfi][qb][0], c, mc)), e1 = __builtin_amdgcn_exp2f(fmaf(sc[kfi][qb][1], c, mc));
                        const float e2 = __builtin_amdgcn_exp2f(fmaf(sc[kfi][qb][2], c, mc)), e3 = __builtin_amdgcn_exp2f(fmaf(sc[kfi][qb][3], c, mc));
                        const fh2_ p01 = __builtin_amdgcn_cvt_pkrtz(e0, e1), p23 = __builtin_amdgcn_cvt_pkrtz(e2, e3);
                        l += ((float)p01[0] + (float)p01[1]) + ((float)p23[0] + (float)p23[1]);      // the denominator sums the SAME fp16 weights that multiply V
                        u[kfi >> 1].h2v[(kfi & 1) * 2] = p01;
                        u[kfi >> 1].h2v[(kfi & 1) * 2 + 1] = p23;
                    }
                    l += __shfl_xor(l, 16, 64);
                    l += __shfl_xor(l, 32, 64);
                    const float inv = 1.f / l;
                    const int row = half * 32 + qb * 16 + l15;
#pragma unroll
                    for (int dv = 0; dv < DV16; ++dv) {
                        f4 o = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[dv][0], u[0].v, z4, 0, 0, 0);
                        o = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[dv][1], u[1].v, o, 0, 0, 0);
                        o = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[dv][2], u[2].v, o, 0, 0, 0);
                        const int dc = dv * 16 + g * 4;
                        if (dc < D) {
                            h4 ov;
#pragma unroll
                            for (int r = 0; r < 4; ++r) ov[r] = (half_t)(o[r] * inv);
                            *reinterpret_cast<h4*>(&T[tile_off(row, h * D + dc)]) = ov;
                        }
                    }
                }
            }
        }
    }
    // ---- phase C: Y = O Wo^T + bias   (the barrier inside project(): all heads of all waves are in the tile)
    A2T(4);
    project(p.Wo_f, true, 5);
    A2T(6);
    // the residual rows of the final pass are requested here, ahead of the two barriers that follow
    constexpr int NIT = BM * CH / 256;
    h8 res[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int id = it * 256 + tid, row = id / CH, c8 = id - row * CH;
        if constexpr (PRE) res[it] = *reinterpret_cast<const h8*>(&Rt[tile_off(row, c8 * 8)]);      // H2 never left the block
        else res[it] = (m0 + row < p.M) ? *reinterpret_cast<const h8*>(p.R + (long)(m0 + row) * p.ldr + c8 * 8) : zero8;
    }
    __syncthreads();                                 // every wave is done reading O: the tile becomes Y
    {
        const int nb = wave * (C / 4) + g * 4;
#pragma unroll
        for (int i = 0; i < NFW; ++i) {
            h4 bv = {0, 0, 0, 0};
            if (p.bias_o) bv = *reinterpret_cast<const h4*>(p.bias_o + nb + i * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                h4 y;
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = (half_t)(acc[i][j][r] + (float)bv[r]);
                *reinterpret_cast<h4*>(&T[tile_off(j * 16 + l15, nb + i * 16)]) = y;
            }
        }
    }
    __syncthreads();
    h8 yv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int id = it * 256 + tid, row = id / CH, c8 = id - row * CH;
        yv[it] = *reinterpret_cast<const h8*>(&T[tile_off(row, c8 * 8)]);
    }
    if (p.stats_out) __syncthreads();                // every thread holds its chunks: the tile may become the statistics scratch
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int id = it * 256 + tid, row = id / CH, c8 = id - row * CH;
        h8 o;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            o[r] = (half_t)((float)yv[it][r] + (float)res[it][r]);
            const float t = (float)o[r];
            s1 += t;
            s2 = fmaf(t, t, s2);
        }
        if (m0 + row < p.M) *reinterpret_cast<h8*>(p.Y + (long)(m0 + row) * p.ldy + c8 * 8) = o;
        if (p.stats_out) scr[id] = float2{s1, s2};
    }
    A2T(7);
    if (p.stats_out) {                               // (sum, sumsq) of the stored values per row and 160-column slot, fixed order
        __syncthreads();
        constexpr int NSLOT = C / 160;
        for (int e = tid; e < BM * NSLOT; e += 256) {
            const int row = e / NSLOT, slot = e - row * NSLOT;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int k = 0; k < 20; ++k) {
                const float2 t = scr[row * CH + slot * 20 + k];
                s1 += t.x;
                s2 += t.y;
            }
            if (m0 + row < p.M) *reinterpret_cast<float2*>(p.stats_out + ((long)(m0 + row) * NSLOT + slot) * 2) = float2{s1, s2};
        }
    }
}

}  // namespace

int uv_launch_frag_pack(const half_t* W, half_t* out, int N, int K, hipStream_t s) {
    UV_REQUIRE(N % 16 == 0 && K % 32 == 0, "frag_pack: N=%d must be a multiple of 16 and K=%d of 32", N, K);
    const long n = (long)(N / 16) * (K / 32) * 64;
    hipLaunchKernelGGL(frag_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, W, out, N, K);
    UV_LAUNCH_CHECK();
    return UV_OK;
}

long uv_attn2_kvf_halfs(int B, int heads, int D) {
    const int KS = (D + 31) / 32, DV16 = (D + 15) / 16;
    return (long)B * heads * (5 * KS + DV16 * 3) * 64 * 8;
}

bool uv_attn2_fused_ok(int C, int heads, int rows_per_branch, int Nkv) {
    static const int env = getenv("UNIVST_ATTN2_FUSED") ? atoi(getenv("UNIVST_ATTN2_FUSED")) : 1;
    return env != 0 && C == 320 && heads == 8 && rows_per_branch % 64 == 0 && Nkv >= 1 && Nkv <= 80;
}

int uv_launch_kv_frag_pack(const half_t* kv, half_t* out, int B, int T, int C, int heads, hipStream_t s) {
    UV_REQUIRE(C % heads == 0 && T >= 1 && T <= 80, "kv_frag_pack: %d keys (<= 80), C=%d, heads=%d", T, C, heads);
    const int D = C / heads;
    if (D == 40) hipLaunchKernelGGL((kv_frag_pack_kernel<40>), dim3(B * heads), dim3(256), 0, s, kv, out, T, C, heads);
    else {
        uv_set_error("kv_frag_pack: head_dim %d not instantiated", D);
        return UV_ERR_UNSUPPORTED;
    }
    UV_LAUNCH_CHECK();
    return UV_OK;
}

int uv_launch_attn2_fused(const Attn2Params& p, int C, hipStream_t s) {
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    UV_REQUIRE(p.M > 0 && uv_attn2_fused_ok(C, p.heads, p.rows_per_branch, p.Nkv), "attn2_fused: C=%d heads=%d rows_per_branch=%d Nkv=%d is not a shape this kernel serves "
               "(C = 320, 8 heads, rows per branch a multiple of 64, <= 80 keys)", C, p.heads, p.rows_per_branch, p.Nkv);
    const bool pre = p.Wp_f != nullptr;
    UV_REQUIRE(p.X && (p.R || pre) && p.Y && p.Wq_f && p.Wo_f && p.kvf && p.ldx % 8 == 0 && p.ldr % 8 == 0 && p.ldy % 8 == 0 && al16(p.X) && al16(p.R) && al16(p.Y) && al16(p.bias_o) &&
               (!p.ln_stats || (p.ln_wsum && p.ln_bias && p.ln_slots > 0)), "attn2_fused: null / misaligned operand");
    UV_REQUIRE(!pre || (p.Rp && p.ldrp % 4 == 0 && (reinterpret_cast<uintptr_t>(p.Rp) & 7) == 0 && (reinterpret_cast<uintptr_t>(p.bias_p) & 7) == 0 && p.ln_wsum && p.ln_bias && !p.ln_stats),
               "attn2_fused: the fused out projection needs its residual rows (8-byte aligned), the folded q weight's wsum / lnb, and no producer statistics");
    const double fl = (pre ? 6.0 : 4.0) * p.M * (double)C * C + 4.0 * p.M * (double)p.Nkv * C;
    uv_prof_begin(UV_CLS_ATTN2_FUSED, fl, 2.0 * (3.0 * p.M * C + 2.0 * C * C), s);
    const dim3 grid((p.M + 63) / 64);
    // (measured and not instantiated: OCC = 3 — three resident blocks per CU at 168 VGPRs, weight ring two deep — 0.190 against 0.182 ms: every phase
    // of a block gets slower (phase A 7.7 k -> 14.6 k cycles per wave), i.e. the kernel is bound by what the blocks of a CU share — weight delivery
    // out of L2, the LDS port, the VALU port of phase B — not by latency a third block could hide)
    if (pre) hipLaunchKernelGGL((attn2_fused_kernel<320, 40, true, 2, true>), grid, dim3(256), 0, s, p);
    else if (p.ln_stats) hipLaunchKernelGGL((attn2_fused_kernel<320, 40, true>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((attn2_fused_kernel<320, 40, false>), grid, dim3(256), 0, s, p);
    uv_prof_end(s);
    UV_LAUNCH_CHECK();
    return UV_OK;
}
