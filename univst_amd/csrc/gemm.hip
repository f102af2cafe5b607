// fp16 MFMA GEMM / implicit-GEMM convolution for gfx950 (CDNA4).
//
//   Y[m][n] = epilogue( sum_k X[m][k] * W[n][k] )          fp16 in, fp32 accumulate, fp16 out
//
// Both operands are K-contiguous ("B^T" layout), which is what v_mfma_f32_16x16x32_f16 wants: every lane
// feeds 8 consecutive k of one row.  The WEIGHT tile is the MFMA A operand and the ACTIVATION tile the B
// operand, so a lane's 4 accumulator registers are 4 consecutive output channels of one output row ->
// 8-byte stores and in-register bias / residual / GEGLU epilogues.
//
// MODE 0: X is a plain [M,K] matrix (Linear layers, attention projections, FF).
// MODE 1: X is gathered on the fly from NHWC activations (implicit GEMM): 3x3 pad-1 conv (stride 1/2),
//         1x1 conv, optional fused nearest x2 upsample of the input, optional channel-concat of two
//         sources (UNet skip connections) — none of these is ever materialised in HBM.
//         K index = tap * (C1 + C2) + c, weights pre-permuted to [Cout][ky][kx][Cin] at load time.
//
// Tile: 128 output rows x (NF*32) output channels x 64 k per step; 4 waves as 2(n) x 2(m), each wave
// NF x 4 fragments of 16x16.  Global -> registers -> LDS staging with the next tile's loads in flight
// under the MFMAs (guide T14); LDS rows padded to 160 B (conflict-free ds_read_b128, see common.h).
// Replaces (reference call sites): resnet.py:64 (PseudoConv3d 2-D conv), attention.py:123,141 (proj_in /
// proj_out), attention.py:375-377,425 + pnp_utils.py:39-43,97 (to_q/k/v/out), diffusers FeedForward/GEGLU,
// diffusers Attention projections, unet_3d_blocks.py:523,618 (torch.cat skip), resnet.py:145 (nearest x2).
#include "common.h"
#include <stdlib.h>

#include "kernels.h"

namespace {

// source of every out-of-range / padding 16-byte piece of the GLDS path (LDS-DMA cannot write immediates)
__device__ __attribute__((aligned(256))) half_t uv_zero_page[128];

// ---- epilogue constants of a tile's 320 output columns, in LDS (gemm_big_kernel / conv_patch_kernel).
// A load placed between two stores of the epilogue waits (vmcnt is one in-order counter on gfx9) for the earlier store's write
// acknowledgement: with the bias fetched inside the store loop every 16-byte store of a K = 320 tile cost a full round trip to L2.
// The per-column constants are therefore fetched ONCE, while the first operand tile is in flight, into a small LDS array next to
// the operand buffers; the store loops read them with ds_read and the only global loads left are the residual rows, requested
// up front.  Layout (floats): [0,320) ln_wsum | [320,640) ln_bias | halfs: bias[320] bias2[320] | float2 (mean, rstd) of 256 rows.
constexpr int EPC_LNB = 320, EPC_HALFS = 640, EPC_ROWST = 960, EPC_TOTAL = 960 + 512;
template <int LNF>
__device__ __forceinline__ void epi_const_stage(const GemmParams& p, int m0, int n0, int bmb, int tid, float* epc) {
    half_t* hb = reinterpret_cast<half_t*>(epc + EPC_HALFS);
    if (tid < 320) {
        const int n = n0 + tid;
        const bool ok = n < p.N;
        hb[tid] = (p.bias && ok) ? p.bias[n] : (half_t)0.f;
        hb[320 + tid] = (p.bias2 && ok) ? p.bias2[n] : (half_t)0.f;
        if (LNF != 2) epc[EPC_LNB + tid] = (p.bias32 && ok) ? p.bias32[(long)(p.w_rows_per_set ? m0 / p.w_rows_per_set : 0) * p.N + n] : 0.f;      // fp32 bias of the tile's weight set
        if (LNF == 2) {
            epc[tid] = ok ? p.ln_wsum[n] : 0.f;
            epc[EPC_LNB + tid] = ok ? p.ln_bias[n] : 0.f;
        }
    }
    if (LNF == 2 && tid < bmb) {     // (mean, rstd) of the tile's rows from the producer's per-slot (sum, sumsq): LayerNorm over K
        float s1 = 0.f, s2 = 0.f;
        if (m0 + tid < p.M) {
            const float2* sp = reinterpret_cast<const float2*>(p.ln_stats) + (long)(m0 + tid) * p.ln_slots;
            for (int e = 0; e < p.ln_slots; ++e) {
                const float2 t = sp[e];
                s1 += t.x;
                s2 += t.y;
            }
        }
        const float inv = 1.f / (float)p.K;
        const float mean = s1 * inv;
        const float var = fmaxf(fmaf(-mean, mean, s2 * inv), 0.f);
        reinterpret_cast<float2*>(epc + EPC_ROWST)[tid] = float2{mean, rsqrtf(var + p.ln_eps)};
    }
}

// Epilogue for ONE output row m: col[i][r] is output channel nb + i*16 + g*4 + r.  bias / per-branch row bias /
// residual / second bias (added after fp16 rounding) / GEGLU on interleaved [16 x | 16 gate] channel blocks.
// cf: the block's LDS constants (null: the 128-row kernels read bias / bias2 from memory), cn = nb - n0.
// LNF (256 x 320 linear kernel only): the LayerNorm that precedes this linear is folded into the epilogue.  With W' = gamma (.) W
// (fp16, made at finalize), wsum[n] = sum_k W'[n][k] and lnb[n] = bias[n] + sum_k beta[k] W[n][k]:
//     LN(x) W^T + bias  =  rstd * (x W'^T  -  mean * wsum)  +  lnb
// so the GEMM runs on the RAW rows x and (mean, rstd) of a row — `ln` — enter only here.
// st (128-row kernels, LNF == 1): += (sum, sum of squares) of the values this lane stores — the row statistics a following folded
// LayerNorm reads; the caller combines the lanes and waves that share the row.
template <int NF, int LNF = 0, bool CL = false>       // CL: per-column constants from LDS (cf), else from memory
__device__ __forceinline__ void gemm_epilogue_row(const GemmParams& p, const f4 (&col)[NF], int m, int nb, int g, const float* cf = nullptr,
                                                  int cn = 0, float2 ln = float2{0.f, 1.f}, float2* st = nullptr) {
    if (m >= p.M) return;
    const half_t* rbias = p.rowbias ? p.rowbias + (long)(m / p.rows_per_rb) * (p.ldrb ? p.ldrb : p.N) : nullptr;
    const half_t* hb = CL ? reinterpret_cast<const half_t*>(cf + EPC_HALFS) + cn : nullptr;
    if (p.geglu) {
        if constexpr (NF % 2 == 0) {
#pragma unroll
            for (int i = 0; i < NF; i += 2) {
                const int nx = nb + i * 16 + g * 4;        // x rows (permuted weight)
                const int ng = nx + 16;                     // gate rows
                if (ng >= p.N) continue;
                const int no = nb / 2 + (i / 2) * 16 + g * 4;
                h4 o, bx = {0, 0, 0, 0}, bg = {0, 0, 0, 0};
                if (LNF == 2) {
                    const float* cc = cf + cn + i * 16 + g * 4;
                    const f4 wx = *reinterpret_cast<const f4*>(cc), wg = *reinterpret_cast<const f4*>(cc + 16);
                    const f4 cx = *reinterpret_cast<const f4*>(cc + EPC_LNB), cg = *reinterpret_cast<const f4*>(cc + EPC_LNB + 16);
                    float xv[4], gv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        xv[r] = fmaf(ln.y, fmaf(-ln.x, wx[r], col[i][r]), cx[r]);
                        gv[r] = fmaf(ln.y, fmaf(-ln.x, wg[r], col[i + 1][r]), cg[r]);
                    }
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        const f2 y = geglu_erf2(f2{xv[r], xv[r + 1]}, f2{gv[r], gv[r + 1]});
                        o[r] = (half_t)y.x;
                        o[r + 1] = (half_t)y.y;
                    }
                    *reinterpret_cast<h4*>(p.Y + (long)m * p.ldy + no) = o;
                    continue;
                }
                if (p.bias) {
                    if constexpr (CL) {
                        bx = *reinterpret_cast<const h4*>(hb + i * 16 + g * 4);
                        bg = *reinterpret_cast<const h4*>(hb + i * 16 + g * 4 + 16);
                    } else {
                        bx = *reinterpret_cast<const h4*>(p.bias + nx);
                        bg = *reinterpret_cast<const h4*>(p.bias + ng);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const f2 y = geglu_erf2(f2{col[i][r] + (float)bx[r], col[i][r + 1] + (float)bx[r + 1]},
                                            f2{col[i + 1][r] + (float)bg[r], col[i + 1][r + 1] + (float)bg[r + 1]});
                    o[r] = (half_t)y.x;
                    o[r + 1] = (half_t)y.y;
                }
                *reinterpret_cast<h4*>(p.Y + (long)m * p.ldy + no) = o;
            }
        }
        return;
    }
    // The residual row is requested before the first STORE: hipcc may not move a load above an earlier store to Y (possible
    // alias), so loads placed inside the store loop each wait out an HBM round trip in turn.
    h4 hres[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        const int n = nb + i * 16 + g * 4;
        hres[i] = (p.R && n + 3 < p.N) ? *reinterpret_cast<const h4*>(p.R + (long)m * p.ldr + n) : h4{0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        const int n = nb + i * 16 + g * 4;
        if (n >= p.N) continue;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = col[i][r];
        if (n + 3 < p.N) {
            if (LNF == 2) {
                f4 ws, cb;
                if constexpr (CL) {
                    ws = *reinterpret_cast<const f4*>(cf + cn + i * 16 + g * 4);
                    cb = *reinterpret_cast<const f4*>(cf + EPC_LNB + cn + i * 16 + g * 4);
                } else {
                    ws = *reinterpret_cast<const f4*>(p.ln_wsum + n);
                    cb = *reinterpret_cast<const f4*>(p.ln_bias + n);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaf(ln.y, fmaf(-ln.x, ws[r], v[r]), cb[r]);
            }
            if (p.bias) {
                h4 bv;
                if constexpr (CL) bv = *reinterpret_cast<const h4*>(hb + i * 16 + g * 4);
                else bv = *reinterpret_cast<const h4*>(p.bias + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += (float)bv[r];
            }
            if (rbias) {
                h4 bv = *reinterpret_cast<const h4*>(rbias + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += (float)bv[r];
            }
            if constexpr (LNF == 3) {          // MM-DiT epilogue (own instantiation: the plain kernels keep their register budget)
                if (p.act) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gelu_tanh_f(v[r]);
                }
                if (p.gate) {
                    const h4 gv = *reinterpret_cast<const h4*>(p.gate + (long)(m / p.rows_per_gate) * p.ld_gate + n);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= (float)gv[r];
                }
            }
            if (p.R) {
                const h4 rv = hres[i];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
            }
            h4 o;
            if (p.bias2) {
                h4 bv;
                if constexpr (CL) bv = *reinterpret_cast<const h4*>(hb + 320 + i * 16 + g * 4);
                else bv = *reinterpret_cast<const h4*>(p.bias2 + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (half_t)((float)(half_t)v[r] + (float)bv[r]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (half_t)v[r];
            }
            if constexpr (LNF == 1 && !CL) {
                if (st) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float t = (float)o[r];
                        st->x += t;
                        st->y = fmaf(t, t, st->y);
                    }
                }
            }
            *reinterpret_cast<h4*>(p.Y + (long)m * p.ldy + n) = o;
        } else {
            for (int r = 0; r < 4 && n + r < p.N; ++r) {
                float t = v[r];
                if (p.bias) t += (float)p.bias[n + r];
                if (rbias) t += (float)rbias[n + r];
                if constexpr (LNF == 3) {
                    if (p.act) t = gelu_tanh_f(t);
                    if (p.gate) t *= (float)p.gate[(long)(m / p.rows_per_gate) * p.ld_gate + n + r];
                }
                if (p.R) t += (float)p.R[(long)m * p.ldr + n + r];
                half_t o = (half_t)t;
                if (p.bias2) o = (half_t)((float)o + (float)p.bias2[n + r]);
                p.Y[(long)m * p.ldy + n + r] = o;
            }
        }
    }
}

// Epilogue of the 256 x 320 tile through LDS: the MFMA fragment layout gives every lane 4 channels of 16 different rows, i.e.
// 32-byte pieces of 16 rows per store (and per residual load).  With one resident block per CU nothing overlaps the
// epilogue, and at K = 320 (the 64x64-level linears) it cost 15-45 us per tile against an 11 us main loop.  Here a wave
// transposes its 64 x 160 fp32 sub-tile 16 rows at a time through a private LDS slab (row stride 164 floats: conflict-free
// ds_write_b128) and reads it back row-major, so residual loads and output stores are 16 B per lane over 320 contiguous bytes.
// Arithmetic order is that of gemm_epilogue_row (fp32: acc + bias + rowbias + residual, one rounding, then bias2).
typedef __fp16 fh2 __attribute__((ext_vector_type(2)));
// sum over the 16 lanes of a DPP row in a fixed order (row_shr 1, 2, 4, 8 with zeros shifted in); the total lands in lane 15 of the row
__device__ __forceinline__ float row16_sum(float v) {
#define UV_ROW_SHR_ADD(n) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + (n), 0xf, 0xf, true))
    UV_ROW_SHR_ADD(1);
    UV_ROW_SHR_ADD(2);
    UV_ROW_SHR_ADD(4);
    UV_ROW_SHR_ADD(8);
#undef UV_ROW_SHR_ADD
    return v;
}
constexpr int EPI_LDW = 164;
// LDS map of the row-statistics producer (bytes; the operand buffers are dead by then): the eight transpose slabs end at
// 8 * 16 * 164 * 4 = 83968; per-wave scratch from 90112 (8 x 2560 B).  The 192-row tile has 128 KB of LDS, the 256-row one 144 KB.
constexpr int EPI_STAT_SCRATCH = 90112;
// cf: the block's LDS constants, cn = nb - n0, rw = first tile row of this wave; scratch: this wave's statistics scratch (LNF).
template <int MJ, int LNF = 0>
__device__ __forceinline__ void gemm_epilogue_lds(const GemmParams& p, const f4 (&acc)[10][MJ], float* slab, int mw, int nb, int lane,
                                                  const float* cf, int cn, int rw = 0, float2* scratch = nullptr, half_t* hs = nullptr) {
    // hs: this wave's 16 x 160 fp16 scratch (5 KB) for the GroupNorm statistics of the stored tile (p.gn_out), else unused
    const int l15 = lane & 15, g = lane >> 4;
    const half_t* hb = reinterpret_cast<const half_t*>(cf + EPC_HALFS) + cn;
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
#pragma unroll
        for (int i = 0; i < 10; ++i) *reinterpret_cast<f4*>(&slab[l15 * EPI_LDW + i * 16 + g * 4]) = acc[i][j];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int mrow0 = mw + j * 16;
        if (!p.geglu) {
            // residual rows of all five passes are requested BEFORE the first store (one hoisted operand: the residual where there
            // is one — conv2, out-projections, FF2 — else the per-branch row bias of conv1; the per-column constants come from LDS)
            h8 rres[5];
            const bool hoist_rb = !p.R && p.rowbias;
            if (p.R || hoist_rb) {
#pragma unroll
                for (int it = 0; it < 5; ++it) {
                    const int id = it * 64 + lane;
                    const int row = id / 20, c = (id - row * 20) * 8;
                    const int m = mrow0 + row < p.M ? mrow0 + row : p.M - 1;
                    const half_t* src = p.R ? p.R + (long)m * p.ldr : p.rowbias + (long)(m / p.rows_per_rb) * (p.ldrb ? p.ldrb : p.N);
                    rres[it] = nb + c < p.N ? *reinterpret_cast<const h8*>(src + nb + c) : h8{0, 0, 0, 0, 0, 0, 0, 0};      // ragged last column tile
                }
            }
#pragma unroll
            for (int it = 0; it < 5; ++it) {
                const int id = it * 64 + lane;
                const int row = id / 20, c = (id - row * 20) * 8;
                const int m = mrow0 + row, n = nb + c;
                const f4 v0 = *reinterpret_cast<const f4*>(&slab[row * EPI_LDW + c]);
                const f4 v1 = *reinterpret_cast<const f4*>(&slab[row * EPI_LDW + c + 4]);
                if (m >= p.M || n >= p.N) continue;            // (n >= N: the ragged last column tile of a linear whose N is not a multiple of 320)
                float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                if (LNF == 2) {
                    const float2 ln = reinterpret_cast<const float2*>(cf + EPC_ROWST)[rw + j * 16 + row];
                    const f4 w0 = *reinterpret_cast<const f4*>(cf + cn + c), w1 = *reinterpret_cast<const f4*>(cf + cn + c + 4);
                    const f4 c0 = *reinterpret_cast<const f4*>(cf + EPC_LNB + cn + c), c1 = *reinterpret_cast<const f4*>(cf + EPC_LNB + cn + c + 4);
                    const float ws[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
                    const float cb[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] = fmaf(ln.y, fmaf(-ln.x, ws[r], v[r]), cb[r]);
                }
                if (p.bias) {
                    const h8 bv = *reinterpret_cast<const h8*>(hb + c);
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] += (float)bv[r];
                }
                if (LNF != 2 && p.bias32) {
                    const f4 b0 = *reinterpret_cast<const f4*>(cf + EPC_LNB + cn + c), b1 = *reinterpret_cast<const f4*>(cf + EPC_LNB + cn + c + 4);
                    v[0] += b0[0]; v[1] += b0[1]; v[2] += b0[2]; v[3] += b0[3];
                    v[4] += b1[0]; v[5] += b1[1]; v[6] += b1[2]; v[7] += b1[3];
                }
                if (p.rowbias) {
                    const h8 bv = hoist_rb ? rres[it] : *reinterpret_cast<const h8*>(p.rowbias + (long)(m / p.rows_per_rb) * (p.ldrb ? p.ldrb : p.N) + n);
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] += (float)bv[r];
                }
                if constexpr (LNF == 3) {
                    if (p.act) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] = gelu_tanh_f(v[r]);
                    }
                    if (p.gate) {
                        const h8 gv = *reinterpret_cast<const h8*>(p.gate + (long)(m / p.rows_per_gate) * p.ld_gate + n);
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] *= (float)gv[r];
                    }
                }
                if (p.R) {
                    const h8 rv = rres[it];
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] += (float)rv[r];
                }
                h8 o;
                if (p.bias2) {
                    const h8 bv = *reinterpret_cast<const h8*>(hb + 320 + c);
#pragma unroll
                    for (int r = 0; r < 8; ++r) o[r] = (half_t)((float)(half_t)v[r] + (float)bv[r]);
                } else {
#pragma unroll
                    for (int r = 0; r < 8; ++r) o[r] = (half_t)v[r];
                }
                *reinterpret_cast<h8*>(p.Y + (long)m * p.ldy + n) = o;
                if (LNF != 1 && p.gn_out) *reinterpret_cast<h8*>(&hs[row * 160 + c]) = o;       // the stored values, row-major, for the group statistics below
                if (LNF == 1) {           // (sum, sum of squares) of the 8 values AS STORED: what the consuming GEMM will read
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const float t = (float)o[r];
                        s1 += t;
                        s2 = fmaf(t, t, s2);
                    }
                    scratch[id] = float2{s1, s2};
                }
            }
            if (LNF == 1) {
                // 16 rows x 20 partials -> one (sum, sumsq) per row and 160-column slot, summed in a fixed order (lane = row*4 + q adds
                // partials 5q..5q+4, then the four q) so the statistics are reproducible
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int row = lane >> 2, q = lane & 3;
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int e = 0; e < 5; ++e) {
                    const float2 t = scratch[row * 20 + q * 5 + e];
                    s1 += t.x;
                    s2 += t.y;
                }
                s1 += __shfl_xor(s1, 1, 64);
                s2 += __shfl_xor(s2, 1, 64);
                s1 += __shfl_xor(s1, 2, 64);
                s2 += __shfl_xor(s2, 2, 64);
                const int m = mrow0 + row;
                if (q == 0 && m < p.M) *reinterpret_cast<float2*>(p.stats_out + ((long)m * (p.N / 160) + nb / 160) * 2) = float2{s1, s2};
            }
            if (LNF != 1 && p.gn_out) {
                // GroupNorm statistics of this 16-row x 160-column piece AS STORED (round 4): lane (r = lane & 15, q = lane >> 4) takes row r,
                // columns 40 q .. 40 q + 39 = 4, 2 or 1 whole channel groups (10, 20, 40 channels per group); sum and sum of squares by
                // v_dot2_f32_f16 (exact fp16 products, fp32 accumulation; two values per instruction, no conversions), then a fixed-order
                // DPP row reduction over the 16 rows.  Lane r = 15 of each q writes (sum, sumsq) of its groups for fragment mrow0 / 16.
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const fh2 one2 = {(__fp16)1.f, (__fp16)1.f};
                union { fh2 h[20]; h8 v[5]; } u;
#pragma unroll
                for (int e = 0; e < 5; ++e) u.v[e] = *reinterpret_cast<const h8*>(&hs[l15 * 160 + g * 40 + e * 8]);
                float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
                const int hp = p.gn_gw >> 1;                       // pairs per group: 5, 10 or 20
#pragma unroll
                for (int k = 0; k < 20; ++k) {
                    const int gi = hp == 5 ? k / 5 : (hp == 10 ? k / 10 : 0);
                    a1[gi] = __builtin_amdgcn_fdot2(u.h[k], one2, a1[gi], false);
                    a2[gi] = __builtin_amdgcn_fdot2(u.h[k], u.h[k], a2[gi], false);
                }
                const int ngpl = 20 / hp;
#pragma unroll
                for (int gi = 0; gi < 4; ++gi) {
                    if (gi < ngpl) {
                        a1[gi] = row16_sum(a1[gi]);
                        a2[gi] = row16_sum(a2[gi]);
                    }
                }
                if (l15 == 15 && mrow0 < p.M) {
                    // layout [sub-group][fragment]: the consumer's reduction walks the fragments of one sub-group — contiguous there
                    float2* dst = reinterpret_cast<float2*>(p.gn_out) + (long)(nb / p.gn_gw + g * ngpl) * (p.M >> 4) + (mrow0 >> 4);
#pragma unroll
                    for (int gi = 0; gi < 4; ++gi)
                        if (gi < ngpl) dst[(long)gi * (p.M >> 4)] = float2{a1[gi], a2[gi]};
                }
            }
        } else {
            // interleaved [16 x | 16 gate] channel blocks -> 80 output columns per row: 16 rows x 10 chunks of 8
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                const int id = it * 64 + lane;
                if (id >= 160) continue;
                const int row = id / 10, cq = id - row * 10;
                const int cx = (cq >> 1) * 32 + (cq & 1) * 8;          // x columns; gates are 16 further
                const int m = mrow0 + row;
                if (m >= p.M) continue;
                const float* sr = &slab[row * EPI_LDW + cx];
                const f4 x0 = *reinterpret_cast<const f4*>(sr), x1 = *reinterpret_cast<const f4*>(sr + 4);
                const f4 g0 = *reinterpret_cast<const f4*>(sr + 16), g1 = *reinterpret_cast<const f4*>(sr + 20);
                float xv[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
                float gv[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
                if (p.bias) {
                    const h8 bx = *reinterpret_cast<const h8*>(hb + cx);
                    const h8 bg = *reinterpret_cast<const h8*>(hb + cx + 16);
#pragma unroll
                    for (int r = 0; r < 8; ++r) { xv[r] += (float)bx[r]; gv[r] += (float)bg[r]; }
                }
                h8 o;
#pragma unroll
                for (int r = 0; r < 8; r += 2) {
                    const f2 y = geglu_erf2(f2{xv[r], xv[r + 1]}, f2{gv[r], gv[r + 1]});
                    o[r] = (half_t)y.x;
                    o[r + 1] = (half_t)y.y;
                }
                *reinterpret_cast<h8*>(p.Y + (long)m * p.ldy + nb / 2 + cq * 8) = o;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}


// GEGLU epilogue of the 256 x 320 tile, round 4.  The register epilogue above stores 8 bytes per lane: one store instruction covers 16
// rows x 32 bytes, so the 80 KB a tile writes reach L2 as 2 560 partial-line requests — an in-kernel ablation (stores predicated off:
// fixed cost per tile 14.0 -> 7.6 us at K = 320; GELU math off: -3.8 us) showed those stores, not the GELU, are the larger half of
// the epilogue.  Here the math stays in the MFMA register layout (all 64 lanes busy, two values per packed instruction), the fp16
// products go through a private 64 x 80 LDS slab per wave (row stride 176 B), and every store instruction writes 16 bytes per lane
// over whole 160-byte row segments: 10 store instructions per wave instead of 20, full 128-byte lines.
constexpr int GEGLU_SLD = 88;       // halfs per slab row (80 + 8 pad: 16-byte aligned rows)
template <int MJ, int LNF>
__device__ __forceinline__ void gemm_epilogue_geglu_slab(const GemmParams& p, const f4 (&acc)[10][MJ], half_t* slab, int mw, int nb, int lane,
                                                         const float* cf, int cn, int rw) {
    const int l15 = lane & 15, g = lane >> 4;
    const half_t* hb = reinterpret_cast<const half_t*>(cf + EPC_HALFS) + cn;
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
        float2 ln = float2{0.f, 1.f};
        if (LNF == 2) ln = reinterpret_cast<const float2*>(cf + EPC_ROWST)[rw + j * 16 + l15];
#pragma unroll
        for (int i = 0; i < 10; i += 2) {
            float xv[4], gv[4];
            if (LNF == 2) {
                const float* cc = cf + cn + i * 16 + g * 4;
                const f4 wx = *reinterpret_cast<const f4*>(cc), wg = *reinterpret_cast<const f4*>(cc + 16);
                const f4 cx = *reinterpret_cast<const f4*>(cc + EPC_LNB), cg = *reinterpret_cast<const f4*>(cc + EPC_LNB + 16);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    xv[r] = fmaf(ln.y, fmaf(-ln.x, wx[r], acc[i][j][r]), cx[r]);
                    gv[r] = fmaf(ln.y, fmaf(-ln.x, wg[r], acc[i + 1][j][r]), cg[r]);
                }
            } else {
                // (no `if (p.bias)` here: epi_const_stage stages zeros for a missing bias, and a branch per 16-column block made every block
                // its own basic block — hipcc then cannot interleave the dependent v_pk_fma_f32 chains of two blocks and pads each link
                // with an s_nop: 423 of them per tile and wave)
                const h4 bx = *reinterpret_cast<const h4*>(hb + i * 16 + g * 4);
                const h4 bg = *reinterpret_cast<const h4*>(hb + i * 16 + g * 4 + 16);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    xv[r] = acc[i][j][r] + (float)bx[r];
                    gv[r] = acc[i + 1][j][r] + (float)bg[r];
                }
            }
            h4 o;
#ifdef UV_GEGLU_X1
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const f2 y = geglu_erf2(f2{xv[r], xv[r + 1]}, f2{gv[r], gv[r + 1]});
                o[r] = (half_t)y.x;
                o[r + 1] = (half_t)y.y;
            }
#else
            f2 y0, y1;
            geglu_erf2x2(f2{xv[0], xv[1]}, f2{gv[0], gv[1]}, f2{xv[2], xv[3]}, f2{gv[2], gv[3]}, y0, y1);
            o[0] = (half_t)y0.x; o[1] = (half_t)y0.y; o[2] = (half_t)y1.x; o[3] = (half_t)y1.y;
#endif
            *reinterpret_cast<h4*>(&slab[(j * 16 + l15) * GEGLU_SLD + (i / 2) * 16 + g * 4]) = o;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    constexpr int NCH = MJ * 16 * 10;              // 16-byte chunks of this wave's 16*MJ rows x 80 columns
#pragma unroll
    for (int it = 0; it < (NCH + 63) / 64; ++it) {
        const int id = it * 64 + lane;
        if (id >= NCH) continue;
        const int row = id / 10, c = id - row * 10;
        const int m = mw + row, n = nb / 2 + c * 8;
        const h8 v = *reinterpret_cast<const h8*>(&slab[row * GEGLU_SLD + c * 8]);
        // (measured and left out: nontemporal stores here are -5 % on the stand-alone K = 320 projection and 0 in the step — the
        // feed-forward's second linear then misses what the plain stores leave in L2 / the Infinity Cache)
        if (m < p.M && n < p.N / 2) *reinterpret_cast<h8*>(p.Y + (long)m * p.ldy + n) = v;
    }
}

// ----------------------------------------------------------------------------------------------------------------
// X-RESIDENT GEGLU projection for K = 320 (round 4: the FF1 of the 64x64 level — the largest single shape of the step).
//
// The 256x320 tile spends 13 of its 23 us at K = 320 outside the k loop (prologue / drain, GELU, store tail) and re-streams the
// activation rows once per column tile.  Here a block owns 128 rows, DMAs their 320 activations into LDS ONCE (80 KB) and walks all
// column tiles of N (256 columns each) with only the weights streaming through two 32 KB buffers — the weight stream never drains
// between column tiles, and the next tile's first k tile is in flight under this tile's epilogue.  Waves 2(m) x 4(n), 64x64 register
// tiles (64 accumulators).  Weight rows are pre-interleaved [x0 x1 g0 g1 | x2 x3 g2 g3 ...] per 16-row fragment
// (geglu_xres_permute_kernel, `GemmParams::geglu == 2`): a lane's accumulator quad is two whole (x, gate) pairs, so the math stays in
// the MFMA register layout, and the fp16 products of a 16-row fragment leave through a 1.25 KB slab per wave as 16-byte pieces.
// Column tile nt, wave column wn, fragment i, lane group g, register r  <->  weight row nt*256 + wn*64 + i*16 + g*4 + r:
//     r < 2: x row of hidden column h = nt*128 + wn*32 + i*8 + g*2 + r;  r >= 2: the gate row of h (r - 2).
// bias / ln_wsum / ln_bias are indexed by that weight row order too.  LNF = 2: the LayerNorm of X folded in (see gemm_epilogue_row).
// Measured stand-alone (tools/probes/xres_probe.hip, M = 196 608, N = 2 560): 0.443 ms = 728 TFLOP/s against 0.556 ms = 580 for the
// 256x320 tile; issuing the previous column tile's epilogue between this tile's MFMAs (a second accumulator set) was slower (0.47 ms).
constexpr int XR_BM = 128, XR_BN = 256, XR_K = 320, XR_NKT = XR_K / 64;
constexpr int XR_XS = XR_NKT * XR_BM * 64, XR_WT = XR_BN * 64, XR_SLAB = 16 * 40;
__host__ __device__ inline int geglu_xres_source_row(int n, int N) {      // which row of the [x | gate] weight is row n of the interleaved one
    const int nt = n / 256, wn = (n % 256) / 64, i = (n % 64) / 16, g = (n % 16) / 4, r = n % 4;
    const int h = nt * 128 + wn * 32 + i * 8 + g * 2 + (r & 1);
    return r < 2 ? h : N / 2 + h;
}
template <int LNF>
__global__ __launch_bounds__(512, 2) void geglu_xres_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) half_t smem[XR_XS + 2 * XR_WT + 8 * XR_SLAB];      // ONE LDS object (see gemm_big_kernel)
    half_t* const Xs = smem;
    half_t* const Wb = smem + XR_XS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int l15 = lane & 15, g = lane >> 4;
    half_t* const slab = smem + XR_XS + 2 * XR_WT + wave * XR_SLAB;
    const int m0 = blockIdx.x * XR_BM;
    const int NT = p.N / XR_BN;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto glds16 = [&](const half_t* src, half_t* dst) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    const int rb = tid >> 3, kc = (tid & 7) ^ (rb & 7);
    // ---- the block's activations, once: k tile kt = 128 rows of 128 B, 16-byte chunks XOR-swizzled with (row & 7)
#pragma unroll
    for (int kt = 0; kt < XR_NKT; ++kt)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + rb + 64 * i;
            glds16(m < p.M ? p.X + (long)m * p.ldx + kt * 64 + kc * 8 : uv_zero_page, Xs + kt * XR_BM * 64 + (64 * i + wave_u * 8) * 64);
        }
    auto issue_w = [&](int nt, int kt, int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(p.W + (long)(nt * XR_BN + rb + 64 * i) * XR_K + kt * 64 + kc * 8, Wb + buf * XR_WT + (64 * i + wave_u * 8) * 64);
    };
    issue_w(0, 0, 0);
    // (mean, rstd) of the lane's four rows (LNF == 2), under the first DMAs
    float2 lnrow[LNF == 2 ? 4 : 1];
    if constexpr (LNF == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + wm * 64 + j * 16 + l15;
            float s1 = 0.f, s2 = 0.f;
            if (m < p.M) {
                const float2* sp = reinterpret_cast<const float2*>(p.ln_stats) + (long)m * p.ln_slots;
                for (int e = 0; e < p.ln_slots; ++e) {
                    const float2 t = sp[e];
                    s1 += t.x;
                    s2 += t.y;
                }
            }
            const float mean = s1 * (1.f / XR_K);
            const float var = fmaxf(fmaf(-mean, mean, s2 * (1.f / XR_K)), 0.f);
            lnrow[j] = float2{mean, rsqrtf(var + p.ln_eps)};
        }
    }
    const int sw = l15 & 7;
    for (int nt = 0; nt < NT; ++nt) {
        // this tile's per-column constants of the lane (weight rows nt*256 + wn*64 + i*16 + g*4 .. +3), requested before the k loop
        const int ncol = nt * XR_BN + wn * 64 + g * 4;
        f4 cw[LNF == 2 ? 4 : 1], cb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (LNF == 2) {
                cw[i] = *reinterpret_cast<const f4*>(p.ln_wsum + ncol + i * 16);
                cb[i] = *reinterpret_cast<const f4*>(p.ln_bias + ncol + i * 16);
            } else {
                h4 b = {0, 0, 0, 0};
                if (p.bias) b = *reinterpret_cast<const h4*>(p.bias + ncol + i * 16);
                cb[i] = f4{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
            }
        }
        f4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < XR_NKT; ++kt) {
            const int step = nt * XR_NKT + kt;
            __syncthreads();                  // vmcnt(0) + barrier: weight tile `step` (at step 0 also the activations) landed; the other buffer is free
            if (kt + 1 < XR_NKT) issue_w(nt, kt + 1, (step + 1) & 1);
            else if (nt + 1 < NT) issue_w(nt + 1, 0, (step + 1) & 1);
            const half_t* Ws = Wb + (step & 1) * XR_WT;
            const half_t* Xk = Xs + kt * XR_BM * 64;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int ch = ((ks * 4 + g) ^ sw) * 8;
                h8 a[4], b[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const h8*>(&Ws[(wn * 64 + i * 16 + l15) * 64 + ch]);
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const h8*>(&Xk[(wm * 64 + j * 16 + l15) * 64 + ch]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
        // ---- epilogue: row fragment j = 16 rows x 32 hidden columns of this wave
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
                f4 va = acc[i][j], vb = acc[i + 1][j];
                if constexpr (LNF == 2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        va[r] = fmaf(lnrow[j].y, fmaf(-lnrow[j].x, cw[i][r], va[r]), cb[i][r]);
                        vb[r] = fmaf(lnrow[j].y, fmaf(-lnrow[j].x, cw[i + 1][r], vb[r]), cb[i + 1][r]);
                    }
                } else {
                    va += cb[i];
                    vb += cb[i + 1];
                }
                f2 ya, yb;
                geglu_erf2x2(f2{va[0], va[1]}, f2{va[2], va[3]}, f2{vb[0], vb[1]}, f2{vb[2], vb[3]}, ya, yb);
                *reinterpret_cast<h2*>(&slab[l15 * 40 + i * 8 + g * 2]) = h2{(half_t)ya.x, (half_t)ya.y};
                *reinterpret_cast<h2*>(&slab[l15 * 40 + (i + 1) * 8 + g * 2]) = h2{(half_t)yb.x, (half_t)yb.y};
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int row = lane >> 2, c = lane & 3;
            const h8 v = *reinterpret_cast<const h8*>(&slab[row * 40 + c * 8]);
            const int m = m0 + wm * 64 + j * 16 + row;
            if (m < p.M) *reinterpret_cast<h8*>(p.Y + (long)m * p.ldy + nt * 128 + wn * 32 + c * 8) = v;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}
// weight / bias rows of a GEGLU projection [x rows | gate rows] -> the interleaved order of geglu_xres_kernel (cols = 1: a bias vector)
__global__ void geglu_xres_permute_kernel(const half_t* __restrict__ in, half_t* __restrict__ out, int rows, int cols) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)rows * cols) return;
    const int c = (int)(i % cols), r = (int)(i / cols);
    out[i] = in[(long)geglu_xres_source_row(r, rows) * cols + c];
}
constexpr int BM_DEFAULT = 128;

// Block tile BM x BN, 256 threads = 4 waves as 2(n) x 2(m); each wave NF x MF MFMA fragments.
// Operand A (weights, LDS [BN][64]) rows n; operand B (activations, LDS [BM][64]) rows m.
// Staging is global_load_lds only: it writes lane-linear 16-byte pieces, so LDS rows are UNPADDED 128 B and the 16-byte
// chunk index is XOR-swizzled with (row & 7) — applied to the per-lane SOURCE address and to the fragment reads.  Two LDS
// buffers, one barrier per 64-wide k tile; the next tile's DMA flies under this tile's MFMAs.  (Register-staged variants of
// this kernel were measured earlier in the round: the ds_write pass + its second barrier cost half the time)
// LNF (linears without split-K): 1 = leaves the row statistics of its output for a following folded LayerNorm (NF = 5 only: the
// block's 160 columns are one slot of GemmParams::stats_out), 2 = folds the LayerNorm of its input into the epilogue (ln_stats; the
// per-column constants come from memory) — what gemm_big_kernel does for the large levels, here for the levels of a frame shard.
template <int NF, int MODE, int MF = 4, int LNF = 0>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmParams p) {
    constexpr int BK = 64, LDSH = 64;
    constexpr int BN = NF * 32;
    constexpr int BM = 32 * MF;          // MF = 4: 128-row tile; MF = 2: 64-row tile for small-M problems that would leave CUs idle
    constexpr int CPR = BK / 8;                            // 16-byte chunks per tile row
    constexpr int RPP = 256 / CPR;                         // rows staged per pass
    constexpr int XL = BM / RPP, WL = (BN + RPP - 1) / RPP;   // staging DMAs per thread
    constexpr int TILE = (BM + BN) * LDSH;
    __shared__ __attribute__((aligned(16))) half_t smem[2 * TILE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wm = wave & 1;
    const int l15 = lane & 15, g = lane >> 4;

    const int nt_n = (p.N + BN - 1) / BN;
    int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int ntiles = nt_n * ((p.M + BM - 1) / BM);
    const int split = lid / ntiles;                     // split-K: blocks of one k range are adjacent (they share W's k range)
    lid -= split * ntiles;
    const int tn = lid % nt_n, tm = lid / nt_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int nk_all = (p.K + BK - 1) / BK;
    const int kt0 = p.splits > 1 ? split * p.ktps : 0;
    const int kt1 = p.splits > 1 ? (kt0 + p.ktps < nk_all ? kt0 + p.ktps : nk_all) : nk_all;

    // ---- per-thread staging geometry: chunk kc (8 halfs) of rows rb + 32*i
    const int rb = tid / CPR;
    const int kc = (tid % CPR) ^ (rb & 7);                 // logical 16-byte chunk this lane stages
    const half_t* xptr[XL];
    int x_iy0[XL], x_ix0[XL], x_img[XL];
    bool x_ok[XL];
#pragma unroll
    for (int i = 0; i < XL; ++i) {
        int m = m0 + rb + RPP * i;
        x_ok[i] = m < p.M;
        if (MODE == 0) {
            xptr[i] = p.X + (long)m * p.ldx + kc * 8;
        } else {
            int hw = p.Ho * p.Wo;
            int img = m / hw, rem = m - img * hw;
            int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            x_iy0[i] = oy * p.stride - p.pady;
            x_ix0[i] = ox * p.stride - p.padx;
            x_img[i] = img * p.Hs;
        }
    }
    const half_t* wptr[WL];
    bool w_ok[WL];
#pragma unroll
    for (int i = 0; i < WL; ++i) {
        int n = n0 + rb + RPP * i;
        w_ok[i] = n < p.N && (rb + RPP * i) < BN;
        wptr[i] = p.W + (long)n * p.K + kc * 8;
    }
    const int Cin = p.C1 + p.C2;
    int tap = 0, cc = kc * 8;          // MODE 1: (tap, channel) of this thread's chunk in the current k tile
    if (MODE == 1) {
        if (p.korder) {
            tap = kt0 % 9;
            cc += (kt0 / 9) * 64;
        } else {
            cc += kt0 * BK;
            tap = cc / Cin;
            cc -= tap * Cin;
        }
    }

    auto advance_k = [&]() {
        if (MODE == 1) {
            if (p.korder) {            // tap-inner: next tap of the same 64-channel slab, then the next slab
                if (++tap == 9) { tap = 0; cc += 64; }
            } else {
                cc += BK;
                while (cc >= Cin) { cc -= Cin; ++tap; }
            }
        }
    };
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto glds16 = [&](const half_t* src, half_t* dst) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    auto issue_tile = [&](int k0, int buf) {      // global -> LDS directly (no VGPR round trip, no ds_write)
        half_t* Xd = smem + buf * TILE + wave_u * 8 * LDSH;
        half_t* Wd = Xd + BM * LDSH;
        const bool kok = (k0 + kc * 8) < p.K;
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < XL; ++i) glds16((x_ok[i] && kok) ? xptr[i] + k0 : uv_zero_page, Xd + RPP * i * LDSH);
        } else {
            const int ky = (p.tapw == 3) ? tap / 3 : tap;        // tapw = taps per kernel row: 3 (3x3), 1 (1x1 and the 3x1 frame conv)
            const int kx = (p.tapw == 3) ? tap - 3 * ky : 0;
            const int He = p.Hs << p.up, We = p.Ws << p.up;
            const bool src2 = cc >= p.C1;
            const half_t* base = src2 ? p.X2 : p.X;
            const int cs = src2 ? p.C2 : p.C1;
            const int co = src2 ? cc - p.C1 : cc;
#pragma unroll
            for (int i = 0; i < XL; ++i) {
                int iy = x_iy0[i] + ky, ix = x_ix0[i] + kx;
                bool ok = x_ok[i] && kok && iy >= 0 && iy < He && ix >= 0 && ix < We;
                long pix = (long)(x_img[i] + (iy >> p.up)) * p.Ws + (ix >> p.up);
                glds16(ok ? base + pix * cs + co : uv_zero_page, Xd + RPP * i * LDSH);
            }
        }
#pragma unroll
        for (int i = 0; i < WL; ++i) glds16((w_ok[i] && kok) ? wptr[i] + k0 : uv_zero_page, Wd + RPP * i * LDSH);
    };

    f4 acc[NF][MF];
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < MF; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    const int nk = kt1 - kt0;            // k tiles of THIS block (all of them unless split-K)
    issue_tile(kt0 * BK, 0);
    advance_k();
    // LNF == 2: (mean, rstd) of the lane's rows from the producer's slots — requested here, under the first tile's DMA
    float2 lnrow[LNF == 2 ? MF : 1];
    if constexpr (LNF == 2) {
#pragma unroll
        for (int j = 0; j < MF; ++j) {
            const int m = m0 + wm * 16 * MF + j * 16 + l15;
            float s1 = 0.f, s2 = 0.f;
            if (m < p.M) {
                const float2* sp = reinterpret_cast<const float2*>(p.ln_stats) + (long)m * p.ln_slots;
                for (int e = 0; e < p.ln_slots; ++e) {
                    const float2 t = sp[e];
                    s1 += t.x;
                    s2 += t.y;
                }
            }
            const float inv = 1.f / (float)p.K;
            const float mean = s1 * inv;
            const float var = fmaxf(fmaf(-mean, mean, s2 * inv), 0.f);
            lnrow[j] = float2{mean, rsqrtf(var + p.ln_eps)};
        }
    }
    const int sw = l15 & 7;
    for (int kt = 0; kt < nk; ++kt) {
        const half_t* Xs = smem + (kt & 1) * TILE;
        const half_t* Ws = Xs + BM * LDSH;
        __syncthreads();                  // (vmcnt(0) + barrier) tile kt landed for every wave; buffer (kt+1)&1 is free
        if (kt + 1 < nk) {                // next tile's DMA flies under this tile's MFMAs
            issue_tile((kt0 + kt + 1) * BK, (kt + 1) & 1);
            advance_k();
        }
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            const int ch = ((ks * 4 + g) ^ sw) * 8;
            h8 a[NF], b[MF];
#pragma unroll
            for (int i = 0; i < NF; ++i) a[i] = *reinterpret_cast<const h8*>(&Ws[(wn * NF * 16 + i * 16 + l15) * LDSH + ch]);
#pragma unroll
            for (int j = 0; j < MF; ++j) b[j] = *reinterpret_cast<const h8*>(&Xs[(wm * 16 * MF + j * 16 + l15) * LDSH + ch]);
#pragma unroll
            for (int i = 0; i < NF; ++i)
#pragma unroll
                for (int j = 0; j < MF; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: lane holds rows n = nb + g*4 + r (r<4) of column m = mb + l15
    if (p.splits > 1) {                   // split-K: raw fp32 partial sums, the epilogue runs in splitk_reduce_kernel
#pragma unroll
        for (int j = 0; j < MF; ++j) {
            const int m = m0 + wm * 16 * MF + j * 16 + l15;
            if (m >= p.M) continue;
            float* row = p.partial + ((long)split * p.M + m) * p.N;
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const int n = n0 + wn * NF * 16 + i * 16 + g * 4;
                if (n + 3 < p.N) *reinterpret_cast<f4*>(row + n) = acc[i][j];
            }
        }
        return;
    }
    if constexpr (LNF == 2) {            // LayerNorm(X) folded in
#pragma unroll
        for (int j = 0; j < MF; ++j) {
            f4 col[NF];
#pragma unroll
            for (int i = 0; i < NF; ++i) col[i] = acc[i][j];
            gemm_epilogue_row<NF, 2>(p, col, m0 + wm * 16 * MF + j * 16 + l15, n0 + wn * NF * 16, g, nullptr, 0, lnrow[j]);
        }
        return;
    }
    if constexpr (LNF == 1) {            // row statistics of the stored outputs: lanes g = 0..3 of a row, then the two wave columns (fixed order)
        static_assert(NF == 5 && MF == 4, "the statistics slot is the block's 160 columns");
        float2* xs = reinterpret_cast<float2*>(smem);      // [128 rows] partials of wave column 1
        __syncthreads();                                    // every wave is done with the operand tiles
#pragma unroll
        for (int j = 0; j < MF; ++j) {
            const int row = wm * 16 * MF + j * 16 + l15;
            f4 col[NF];
#pragma unroll
            for (int i = 0; i < NF; ++i) col[i] = acc[i][j];
            float2 st = float2{0.f, 0.f};
            gemm_epilogue_row<NF, 1>(p, col, m0 + row, n0 + wn * NF * 16, g, nullptr, 0, float2{0.f, 1.f}, &st);
            st.x += __shfl_xor(st.x, 16, 64);
            st.y += __shfl_xor(st.y, 16, 64);
            st.x += __shfl_xor(st.x, 32, 64);
            st.y += __shfl_xor(st.y, 32, 64);
            if (wn == 1 && g == 0) xs[row] = st;
            acc[0][j][0] = st.x;                            // kept for the second pass (wave column 0)
            acc[0][j][1] = st.y;
        }
        __syncthreads();
        if (wn == 0 && g == 0) {
#pragma unroll
            for (int j = 0; j < MF; ++j) {
                const int row = wm * 16 * MF + j * 16 + l15;
                const int m = m0 + row;
                if (m < p.M) {
                    const float2 o = xs[row];
                    *reinterpret_cast<float2*>(p.stats_out + ((long)m * (p.N / 160) + n0 / 160) * 2) = float2{acc[0][j][0] + o.x, acc[0][j][1] + o.y};
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < MF; ++j) {
        f4 col[NF];
#pragma unroll
        for (int i = 0; i < NF; ++i) col[i] = acc[i][j];
        if (p.act || p.gate) gemm_epilogue_row<NF, 3>(p, col, m0 + wm * 16 * MF + j * 16 + l15, n0 + wn * NF * 16, g);
        else gemm_epilogue_row<NF>(p, col, m0 + wm * 16 * MF + j * 16 + l15, n0 + wn * NF * 16, g);
    }
}

// split-K second pass: sum the fp32 partials in split order (deterministic) and run the normal epilogue on 4 channels.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmParams p) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int n4 = p.N / 4;
    if (i >= (long)p.M * n4) return;
    const int m = (int)(i / n4), n = (int)(i - (long)m * n4) * 4;
    f4 a = *reinterpret_cast<const f4*>(p.partial + (long)m * p.N + n);
    for (int sidx = 1; sidx < p.splits; ++sidx) a += *reinterpret_cast<const f4*>(p.partial + ((long)sidx * p.M + m) * p.N + n);
    f4 col[1] = {a};
    if (p.act || p.gate) gemm_epilogue_row<1, 3>(p, col, m, n, 0);
    else gemm_epilogue_row<1>(p, col, m, n, 0);
}


// ----------------------------------------------------------------------------------------------------------------
// Large-M kernel: 256 x 320 output tile, 8 waves as 4(m) x 2(n), each wave 64(m) x 160(n) = 4 x 10 fragments
// (160 fp32 accumulators).  Every SD-v1.5 layer width is a multiple of 320, so there is no tile waste; the
// operand traffic per flop is 2.2x lower than the 128x128 kernel, which is what lifts the L2-bandwidth ceiling
// those tiles sit on.  Staging is global_load_lds only (two 72 KB LDS buffers, unpadded 128 B
// rows, XOR-swizzled chunks), one barrier per 64-wide k tile, 80 MFMAs per wave between barriers.
// LNF: 0 plain, 1 emits the row statistics of its output (GemmParams::stats_out), 2 folds a LayerNorm of its input (ln_stats),
// 3 the MM-DiT epilogue (GemmParams::act / gate)
template <int MODE, int MJ, int LNF = 0>
__global__ __launch_bounds__(512, 2) void gemm_big_kernel(GemmParams p) {
    // MJ = 16-row fragments per wave along M: 4 -> 256-row tile, 3 -> 192-row tile (same kernel, chosen per problem so that
    // the tile count fills whole rounds of the 256 CUs: 49152 and 12288 rows are 256 / 64 tiles of 192)
    constexpr int NF = 10, BMB = 64 * MJ, BNB = 320, BK = 64, LDSH = 64;
    constexpr int TILE = (BMB + BNB) * LDSH;                 // halfs per buffer (72 KB)
    // ONE LDS object on purpose: with a second __shared__ array hipcc keeps pointer provenance on every LDS access and then puts a
    // vmcnt(0) in front of the fragment reads of each k tile ("may alias the LDS-DMA in flight"), which serialises DMA and MFMAs
    // (-25 % on the long-K linears).  The epilogue constants (epi_const_stage) live behind the two operand buffers.
    __shared__ __attribute__((aligned(16))) half_t smem[2 * TILE + 2 * EPC_TOTAL];
    float* const epc = reinterpret_cast<float*>(smem + 2 * TILE);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wn = wave & 1, wm = wave >> 1;
    const int l15 = lane & 15, g = lane >> 4;
    const int nt_n = (p.N + BNB - 1) / BNB;
    int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int ntiles = nt_n * ((p.M + BMB - 1) / BMB);
    const int split = lid / ntiles;                           // split-K (see gemm_kernel)
    lid -= split * ntiles;
    // Tile order.  The ~32 tiles an XCD runs at a time should share operands that fit its 4 MB L2.  With the column tile as the
    // fastest index and MANY column tiles (GEGLU / QKV projections of the 32x32 and 16x16 levels: 6 .. 32 of them, weights of 6 - 26
    // MB), 32 concurrent tiles are 1 - 2 row tiles x all of W: W streams from HBM once per pair of row tiles (1.7 GB per launch of
    // the 32x32-level GEGLU projection).  p.tile_gn > 0: groups of tile_gm row tiles x tile_gn column tiles; the column blocks of
    // one row group run back to back (its activation rows stay in L2), W is read once per row group.  Measured: -2..4 % on the
    // GEGLU / QKV projections of the 32x32 and 16x16 levels — they were closer to the matrix pipe's limit than to the memory's.
    int tn, tm;
    if (p.tile_gn > 0) {
        const int nt_m = ntiles / nt_n;
        const int mg = lid / (p.tile_gm * nt_n);                          // row group
        const int gm = nt_m - mg * p.tile_gm < p.tile_gm ? nt_m - mg * p.tile_gm : p.tile_gm;      // its height (the last one may be short)
        const int rem = lid - mg * p.tile_gm * nt_n;
        const int nb = rem / (gm * p.tile_gn), r2 = rem - nb * gm * p.tile_gn;
        tm = mg * p.tile_gm + r2 % gm;
        tn = nb * p.tile_gn + r2 / gm;
    } else {
        tn = lid % nt_n;
        tm = lid / nt_n;
    }
    const int m0 = tm * BMB, n0 = tn * BNB;
    const int nk_all = (p.K + BK - 1) / BK;
    const int kt0 = p.splits > 1 ? split * p.ktps : 0;
    const int kt1 = p.splits > 1 ? (kt0 + p.ktps < nk_all ? kt0 + p.ktps : nk_all) : nk_all;

    const int rb = tid >> 3;                                  // 0..63
    const int kc = (tid & 7) ^ (rb & 7);                      // logical chunk staged by this lane
    unsigned xoff[MJ];
    int x_iy0[MJ], x_ix0[MJ], x_img[MJ];
    bool x_ok[MJ];
#pragma unroll
    for (int i = 0; i < MJ; ++i) {
        int m = m0 + rb + 64 * i;
        x_ok[i] = m < p.M;
        if (MODE == 0) {
            xoff[i] = (unsigned)((long)m * p.ldx + kc * 8);
        } else {
            int hw = p.Ho * p.Wo;
            int img = m / hw, rem = m - img * hw;
            int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            x_iy0[i] = oy * p.stride - p.pady;
            x_ix0[i] = ox * p.stride - p.padx;
            x_img[i] = img * p.Hs;
        }
    }
    // Fast im2col addressing for the common case (3x3, tap-inner k order, no fused upsample): the pixel offset of tap
    // (ky, kx) is the row's own base plus a wave-uniform delta, and padding validity is a 9-bit mask per row computed
    // once — 3 VALU per DMA instead of ~25 (the conv kernel issued 149 non-MFMA VALU per 80 MFMAs; tools/pmc_gemm.sh).
    const bool fast_conv = MODE == 1 && p.korder && p.up == 0 && p.taps == 9;      // (korder implies the symmetric 3x3: launcher)
    int x_pix0[MJ];
    unsigned x_mask[MJ];
    if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < MJ; ++i) {
            x_pix0[i] = (x_img[i] + x_iy0[i]) * p.Ws + x_ix0[i];
            unsigned mk = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int iy = x_iy0[i] + t / 3, ix = x_ix0[i] + t % 3;
                if (x_ok[i] && iy >= 0 && iy < p.Hs && ix >= 0 && ix < p.Ws) mk |= 1u << t;
            }
            x_mask[i] = mk;
        }
    }
    unsigned woff[5];
    bool w_ok[5];
    const long wset = (MODE == 0 && p.w_rows_per_set) ? (long)(m0 / p.w_rows_per_set) * p.N * p.K : 0;      // this tile's weight set (GemmParams::w_rows_per_set)
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        int n = n0 + rb + 64 * i;
        w_ok[i] = n < p.N;
        woff[i] = (unsigned)(wset + (long)n * p.K + kc * 8);
    }
    const int Cin = p.C1 + p.C2;
    int tap = 0, cc = kc * 8;
    if (MODE == 1) {
        if (p.korder) {
            tap = kt0 % 9;
            cc += (kt0 / 9) * 64;
        } else {
            cc += kt0 * BK;
            tap = cc / Cin;
            cc -= tap * Cin;
        }
    }
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const half_t* zp = uv_zero_page;
    auto glds16 = [&](const half_t* src, half_t* dst) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    auto issue_tile = [&](int k0, int buf, int parts = 3) {      // parts: 1 = activation rows, 2 = weight rows (+ k-state advance)
        half_t* Xd = smem + buf * TILE + wave_u * 8 * LDSH;
        half_t* Wd = Xd + BMB * LDSH;
        const bool kok = (k0 + kc * 8) < p.K;
        if (!(parts & 1)) {
        } else if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < MJ; ++i) glds16((x_ok[i] && kok) ? p.X + xoff[i] + k0 : zp, Xd + 64 * i * LDSH);
        } else if (fast_conv) {
            const int tap_u = __builtin_amdgcn_readfirstlane(tap);             // (tap, slab) are block-uniform in this k order
            const int slab_u = __builtin_amdgcn_readfirstlane(cc - kc * 8);
            const bool src2 = slab_u >= p.C1;
            const half_t* base = src2 ? p.X2 : p.X;
            const int cs = src2 ? p.C2 : p.C1;
            const int ky = tap_u / 3, kx = tap_u - 3 * ky;
            const long delta = (long)(ky * p.Ws + kx) * cs + (src2 ? slab_u - p.C1 : slab_u) + kc * 8;
#pragma unroll
            for (int i = 0; i < MJ; ++i) {
                const bool ok = kok && ((x_mask[i] >> tap_u) & 1u);
                glds16(ok ? base + (long)x_pix0[i] * cs + delta : zp, Xd + 64 * i * LDSH);
            }
        } else {
            const int ky = (p.tapw == 3) ? tap / 3 : tap;        // tapw = taps per kernel row: 3 (3x3), 1 (1x1 and the 3x1 frame conv)
            const int kx = (p.tapw == 3) ? tap - 3 * ky : 0;
            const int He = p.Hs << p.up, We = p.Ws << p.up;
            const bool src2 = cc >= p.C1;
            const half_t* base = src2 ? p.X2 : p.X;
            const int cs = src2 ? p.C2 : p.C1;
            const int co = src2 ? cc - p.C1 : cc;
#pragma unroll
            for (int i = 0; i < MJ; ++i) {
                int iy = x_iy0[i] + ky, ix = x_ix0[i] + kx;
                bool ok = x_ok[i] && kok && iy >= 0 && iy < He && ix >= 0 && ix < We;
                long pix = (long)(x_img[i] + (iy >> p.up)) * p.Ws + (ix >> p.up);
                glds16(ok ? base + pix * cs + co : zp, Xd + 64 * i * LDSH);
            }
        }
        if (!(parts & 2)) return;
#pragma unroll
        for (int i = 0; i < 5; ++i) glds16((w_ok[i] && kok) ? p.W + woff[i] + k0 : zp, Wd + 64 * i * LDSH);
        if (MODE == 1) {
            if (p.korder) {
                if (++tap == 9) { tap = 0; cc += 64; }
            } else {
                cc += BK;
                while (cc >= Cin) { cc -= Cin; ++tap; }
            }
        }
    };

    const int nk = kt1 - kt0;
    issue_tile(kt0 * BK, 0);
    if (p.splits <= 1) epi_const_stage<LNF>(p, m0, n0, BMB, tid, epc);      // lands with the first tile, visible after the first barrier
    f4 acc[NF][MJ];
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    const int sw = l15 & 7;
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();                  // vmcnt(0) + barrier: tile kt landed everywhere, buffer (kt+1)&1 is free
        const bool more = kt + 1 < nk;
        if (more) issue_tile((kt0 + kt + 1) * BK, (kt + 1) & 1, 1);      // activation rows now, weight rows before the second k-half
        const half_t* Xs = smem + (kt & 1) * TILE;
        const half_t* Ws = Xs + BMB * LDSH;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks == 1 && more) issue_tile((kt0 + kt + 1) * BK, (kt + 1) & 1, 2);
            const int ch = ((ks * 4 + g) ^ sw) * 8;
            h8 a[NF];
#pragma unroll
            for (int i = 0; i < NF; ++i) a[i] = *reinterpret_cast<const h8*>(&Ws[(wn * 160 + i * 16 + l15) * LDSH + ch]);
#pragma unroll
            for (int j = 0; j < MJ; ++j) {
                h8 b = *reinterpret_cast<const h8*>(&Xs[(wm * 16 * MJ + j * 16 + l15) * LDSH + ch]);
#pragma unroll
                for (int i = 0; i < NF; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b, acc[i][j], 0, 0, 0);
            }
        }
    }
    if (p.splits > 1) {                   // split-K: raw fp32 partials; splitk_reduce_kernel runs the epilogue
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
            const int m = m0 + wm * 16 * MJ + j * 16 + l15;
            if (m >= p.M) continue;
            float* row = p.partial + ((long)split * p.M + m) * p.N + n0 + wn * 160 + g * 4;
#pragma unroll
            for (int i = 0; i < NF; ++i) *reinterpret_cast<f4*>(row + i * 16) = acc[i][j];
        }
        return;
    }
    if (p.epi_lds == 3) {                 // GEGLU: math in registers, fp16 products through a per-wave LDS slab, 16-byte row-contiguous stores
        if constexpr (MODE == 0 && (LNF == 0 || LNF == 2)) {
            __syncthreads();              // every wave is done with the operand tiles
            gemm_epilogue_geglu_slab<MJ, LNF>(p, acc, smem + wave * (64 * GEGLU_SLD), m0 + wm * 16 * MJ, n0 + wn * 160, lane, epc, wn * 160, wm * 16 * MJ);
        }
        return;
    }
    if (p.epi_lds) {
        __syncthreads();                  // every wave is done with the operand tiles: smem becomes the transpose scratch
        gemm_epilogue_lds<MJ, LNF>(p, acc, reinterpret_cast<float*>(smem) + wave * (16 * EPI_LDW), m0 + wm * 16 * MJ, n0 + wn * 160, lane, epc, wn * 160,
                                   wm * 16 * MJ, reinterpret_cast<float2*>(reinterpret_cast<char*>(smem) + EPI_STAT_SCRATCH) + wave * 320,
                                   reinterpret_cast<half_t*>(reinterpret_cast<char*>(smem) + EPI_STAT_SCRATCH) + wave * 2560);
        return;
    }
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
        f4 col[NF];
#pragma unroll
        for (int i = 0; i < NF; ++i) col[i] = acc[i][j];
        gemm_epilogue_row<NF, LNF, true>(p, col, m0 + wm * 16 * MJ + j * 16 + l15, n0 + wn * 160, g, epc, wn * 160,
                                   (LNF == 2) ? reinterpret_cast<const float2*>(epc + EPC_ROWST)[wm * 16 * MJ + j * 16 + l15] : float2{0.f, 1.f});
    }
}


// ----------------------------------------------------------------------------------------------------------------
// 3x3 / stride-1 convolution (optionally over the nearest-x2 upsampled input) from an INPUT PATCH kept in LDS (the 64x64 ..
// 16x16 levels: 19 of the step's 46.7 TFLOP).
//
// gemm_big_kernel<1> stages one im2col tile per k tile: every input pixel enters LDS nine times (once per tap), the
// LDS-DMA lands at ~64 B/clk/CU, and its 72 KB per k tile next to 224 KB of fragment reads make the LDS port the
// co-limiter of the tile.  Here the block's output tile is R whole image rows; for a 32-channel slab the
// (R+2) x (W+2) input pixels those rows touch are DMA'd ONCE (25 KB at the 64x64 level) and the nine taps read their B
// fragments from it at shifted pixel offsets — no per-tap im2col addressing, no padding masks (halo pixels are zero in
// the patch), 37 % fewer LDS-DMA bytes.  Weights stream as before (320 x 64 k per barrier, two buffers); a k tile is two
// consecutive (slab, tap) units of the k order [Cin/32][9][32] (GemmParams::W32), so one barrier still covers 80 MFMAs
// per wave; patches sit in a two-slab ring and are refilled two slabs ahead.
//
// Patch image: pixel-major, 64 B per pixel (32 channels), 16-byte chunk c of pixel p stored at chunk c ^ (((p >> 2) & 1) << 1):
// conflict-free for the ds_read_b128 lane groups at ANY pixel shift (brute-forced against MI355X_MICROARCH.md §LDS).
// The swizzle is applied to the per-lane SOURCE address (the DMA writes lane-linear) and to the fragment reads.
// A tile may straddle image boundaries (192-row tiles do; at the 8x8 level a tile holds three or four whole images): ONE zero
// row is inserted between two images, which serves as the bottom padding of the upper image and the top padding of the lower
// one.  Image widths below 16 are fine as long as they divide 16 (a 16-pixel MFMA fragment then covers whole image rows).
template <int MJ>
__global__ __launch_bounds__(512, 2) void conv_patch_kernel(GemmParams p) {
    constexpr int NF = 10, BMB = 64 * MJ, BNB = 320, LDSH = 64;
    constexpr int PROUNDS = 4;                               // DMA rounds per patch (512 lanes x 16 B each)
    constexpr int PBUF = PROUNDS * 512 * 8;                  // halfs per patch buffer (32 KB)
    constexpr int WT = BNB * LDSH;                           // halfs per weight buffer (40 KB)
    __shared__ __attribute__((aligned(16))) half_t smem[2 * PBUF + 2 * WT + 2 * EPC_TOTAL];     // 144 KB + the epilogue constants (one LDS object: see gemm_big_kernel)
    float* const epc = reinterpret_cast<float*>(smem + 2 * PBUF + 2 * WT);
    half_t* const Pb = smem;
    half_t* const Wb = smem + 2 * PBUF;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wn = wave & 1, wm = wave >> 1;
    const int l15 = lane & 15, g = lane >> 4;
    const int nt_n = p.N / BNB;
    int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int ntiles = nt_n * ((p.M + BMB - 1) / BMB);
    const int split = lid / ntiles;                          // split-K over 64-channel slab pairs (p.ktps = 9 * pairs per split)
    lid -= split * ntiles;
    const int tn = lid % nt_n, tm = lid / nt_n;
    const int m0 = tm * BMB, n0 = tn * BNB;
    const int Wd = p.Wo, PW = Wd + 2, Himg = p.Ho;
    const int rows_total = p.M / Wd;                         // image rows in the stack of all images
    const int R = BMB / Wd;
    const int gr0 = m0 / Wd;                                 // first image row of the tile in the stack of all images
    // patch rows: [row above the tile | tile rows, with one zero row wherever two images meet | row below the tile].  In the
    // sequence "image rows + one separator per image" (period Himg + 1) the tile's first row sits at position r0, so patch row
    // py is sequence element py - 1 + r0; elements at position Himg of a period (and the one before the first image) are zero rows.
    const int r0 = gr0 % Himg;
    const int nsep = (gr0 + R - 1) / Himg - gr0 / Himg;      // image boundaries strictly inside the tile
    const int nprow = R + 2 + nsep;
    const int npieces = nprow * PW * 4;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const half_t* zp = uv_zero_page;
    auto glds16 = [&](const half_t* src, half_t* dst) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };

    // ---- patch staging geometry (independent of the slab): source pixel index (-1: zero) and logical chunk per round
    int ppix[PROUNDS], pch[PROUNDS];
#pragma unroll
    for (int i = 0; i < PROUNDS; ++i) {
        const int q = i * 512 + tid;
        const int pp = q >> 2, cst = q & 3;
        const int py = pp / PW, px = pp - py * PW;
        const int sq = py - 1 + r0;                                      // element of the rows-plus-separators sequence
        const int seg = sq >= 0 ? sq / (Himg + 1) : 0, pos = sq >= 0 ? sq - seg * (Himg + 1) : Himg;
        const int gr = gr0 - r0 + seg * Himg + pos;
        const bool ok = q < npieces && px >= 1 && px <= Wd && pos != Himg && gr < rows_total;      // pos == Himg: zero row (padding)
        const int img = gr / Himg, yy = gr - img * Himg;                 // fused nearest x2 upsample (p.up): source pixel = (y >> 1, x >> 1)
        ppix[i] = ok ? (img * p.Hs + (yy >> p.up)) * p.Ws + ((px - 1) >> p.up) : -1;
        pch[i] = (cst ^ (((pp >> 2) & 1) << 1)) * 8;
    }
    auto issue_patch = [&](int slab) {
        const int c0 = slab * 32;
        const bool src2 = c0 >= p.C1;
        const half_t* base = src2 ? p.X2 : p.X;
        const int cs = src2 ? p.C2 : p.C1;
        const int co = src2 ? c0 - p.C1 : c0;
        half_t* dst = Pb + (slab & 1) * PBUF + wave_u * 64 * 8;
#pragma unroll
        for (int i = 0; i < PROUNDS; ++i)
            if (i * 512 < npieces) glds16(ppix[i] >= 0 ? base + (long)ppix[i] * cs + co + pch[i] : zp, dst + i * 512 * 8);
    };
    // ---- weight staging (as gemm_big_kernel): rows rb + 64 i, chunk kc, XOR-swizzled with the row
    const int rb = tid >> 3;
    const int kc = (tid & 7) ^ (rb & 7);
    unsigned woff[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) woff[i] = (unsigned)((long)(n0 + rb + 64 * i) * p.K + kc * 8);
    auto issue_w = [&](int t) {
        half_t* dst = Wb + (t & 1) * WT + wave_u * 8 * LDSH;
#pragma unroll
        for (int i = 0; i < 5; ++i) glds16(p.W32 + woff[i] + t * 64, dst + 64 * i * LDSH);
    };

    // ---- fragment geometry: patch pixel (tap 0,0) of this lane's row in each of the wave's MJ fragments
    int pbase[MJ];
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
        const int ml = wm * 16 * MJ + j * 16 + l15;          // this lane's output pixel in the tile
        const int ry = ml / Wd, rx = ml - ry * Wd;
        const int seps = (gr0 + ry) / Himg - gr0 / Himg;     // zero rows inserted above it
        pbase[j] = (ry + seps) * PW + rx;
    }

    const int nslab_all = (p.C1 + p.C2) / 32;
    const int s_begin = p.splits > 1 ? split * (p.ktps / 9) * 2 : 0;
    const int s_end = p.splits > 1 ? (s_begin + (p.ktps / 9) * 2 < nslab_all ? s_begin + (p.ktps / 9) * 2 : nslab_all) : nslab_all;
    const int nslab = s_end - s_begin;                       // slabs of THIS block (an even number)
    const int T = nslab * 9 / 2;
    const int t_off = s_begin * 9 / 2;                       // first k tile of this block in the weight's k order
    issue_patch(s_begin);
    if (nslab > 1) issue_patch(s_begin + 1);
    issue_w(t_off);
    if (p.splits <= 1) epi_const_stage<0>(p, m0, n0, BMB, tid, epc);
    f4 acc[NF][MJ];
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    int next_patch = 2;                                      // next slab to stage, relative to s_begin
    int u_slab = s_begin, u_tap = 0;
    const int sw = l15 & 7;
    for (int t = 0; t < T; ++t) {
        __syncthreads();                  // vmcnt(0) + barrier: everything issued so far landed; tile t-1 is consumed everywhere
        if (next_patch < nslab && 9 * (next_patch - 1) <= 2 * t) {     // slab next_patch-2 is fully consumed: refill its buffer
            issue_patch(s_begin + next_patch);
            ++next_patch;
        }
        const half_t* Ws = Wb + ((t_off + t) & 1) * WT;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks == 1 && t + 1 < T) issue_w(t_off + t + 1);                // next weight tile: before the second k-half (measured better than before the first)
            const half_t* Ps = Pb + (u_slab & 1) * PBUF;
            const int ky = u_tap / 3, kx = u_tap - 3 * ky;
            const int delta = ky * PW + kx;
            const int ch = ((ks * 4 + g) ^ sw) * 8;
            h8 a[NF];
#pragma unroll
            for (int i = 0; i < NF; ++i) a[i] = *reinterpret_cast<const h8*>(&Ws[(wn * 160 + i * 16 + l15) * LDSH + ch]);
#pragma unroll
            for (int j = 0; j < MJ; ++j) {
                const int pp = pbase[j] + delta;
                const h8 b = *reinterpret_cast<const h8*>(&Ps[pp * 32 + ((g ^ (((pp >> 2) & 1) << 1)) << 3)]);
#pragma unroll
                for (int i = 0; i < NF; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b, acc[i][j], 0, 0, 0);
            }
            if (++u_tap == 9) { u_tap = 0; ++u_slab; }
        }
    }
    if (p.splits > 1) {                   // split-K: raw fp32 partials; splitk_reduce_kernel runs the epilogue
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
            const int m = m0 + wm * 16 * MJ + j * 16 + l15;
            if (m >= p.M) continue;
            float* row = p.partial + ((long)split * p.M + m) * p.N + n0 + wn * 160 + g * 4;
#pragma unroll
            for (int i = 0; i < NF; ++i) *reinterpret_cast<f4*>(row + i * 16) = acc[i][j];
        }
        return;
    }
    if (p.epi_lds) {
        __syncthreads();
        gemm_epilogue_lds<MJ>(p, acc, reinterpret_cast<float*>(smem) + wave * (16 * EPI_LDW), m0 + wm * 16 * MJ, n0 + wn * 160, lane, epc, wn * 160, 0, nullptr,
                              reinterpret_cast<half_t*>(reinterpret_cast<char*>(smem) + EPI_STAT_SCRATCH) + wave * 2560);
        return;
    }
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
        f4 col[NF];
#pragma unroll
        for (int i = 0; i < NF; ++i) col[i] = acc[i][j];
        gemm_epilogue_row<NF, 0, true>(p, col, m0 + wm * 16 * MJ + j * 16 + l15, n0 + wn * 160, g, epc, wn * 160);
    }
}


// y[m][n] = sum_k act(x[m][k]) W[n][k] + b[n], M <= 8: one wave per output column (time embeddings).
__global__ __launch_bounds__(256) void linear_small_kernel(const half_t* __restrict__ x, const half_t* __restrict__ W,
                                                           const half_t* __restrict__ b, half_t* __restrict__ y,
                                                           int M, int N, int K, int silu_in) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float acc[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m] = 0.f;
    for (int k = lane * 8; k < K; k += 64 * 8) {
        h8 w = *reinterpret_cast<const h8*>(W + (long)n * K + k);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (m < M) {
                h8 xv = *reinterpret_cast<const h8*>(x + (long)m * K + k);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float xf = (float)xv[e];
                    if (silu_in) xf = silu_f(xf);
                    acc[m] += xf * (float)w[e];
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        float s = wave_sum(acc[m]);
        if (lane == 0 && m < M) y[(long)m * N + n] = (half_t)(s + (b ? (float)b[n] : 0.f));
    }
}

}  // namespace

static int uv_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        n = 256;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0)
            n = pr.multiProcessorCount;
    }
    return n;
}

// Split-K factor for `ntiles` output tiles, `nk` k tiles each, on `slots` concurrently resident blocks: minimise
//   ceil(ntiles * s / slots) rounds x ceil(nk / s) k tiles (us_per_ktile each)  +  the fp32 partial traffic (s writes + s reads)
// — rounds, not raw block counts: 48 tiles x 6 splits = 288 blocks is TWO rounds on 256 CUs, 48 x 5 = 240 is one.
static int uv_pick_splits(long ntiles, int nk, long slots, int min_ktps, int max_s, double us_per_ktile, double mn_bytes) {
    int best_s = 1;
    double best = (double)((ntiles + slots - 1) / slots) * nk * us_per_ktile;
    for (int s = 2; s <= max_s; ++s) {
        const int ktps = (nk + s - 1) / s;
        if (ktps < min_ktps) break;
        const double t = (double)((ntiles * s + slots - 1) / slots) * ktps * us_per_ktile + 2.0 * s * mn_bytes / 4.0e6;   // 4 TB/s = 4e6 B/us
        if (t < best * 0.95) { best = t; best_s = s; }
    }
    return best_s;
}

// LDS-patch 3x3 conv (conv_patch_kernel): whole image rows per tile, 16-pixel fragments inside one image row, patch <= 512 pixels
static bool uv_conv_patch_eligible(const GemmParams& p, int bmb) {
    return p.W32 && p.taps == 9 && p.stride == 1 && p.C1 % 32 == 0 && p.C2 % 32 == 0 && (p.C1 + p.C2) % 64 == 0 && p.N % 320 == 0 && (p.Wo % 16 == 0 || 16 % p.Wo == 0) &&
           bmb % p.Wo == 0 && (bmb / p.Wo + 2 + bmb / p.Wo / p.Ho + 1) * (p.Wo + 2) <= 512 && !p.geglu;
}

// ONE copy of what the 256x320 path requires of a problem's shape / environment: used by the launcher below and by the predicate the
// UNet graph consults before it decides to fold a LayerNorm (the two used to be separate copies that could drift apart).
struct BigEnv {
    int nobig, epi;
    long bigmin;
};
static const BigEnv& big_env() {
    static const BigEnv e = {getenv("UNIVST_GEMM_NOBIG") ? atoi(getenv("UNIVST_GEMM_NOBIG")) : 0,
                             getenv("UNIVST_GEMM_EPI") ? atoi(getenv("UNIVST_GEMM_EPI")) : 1,
                             (getenv("UNIVST_GEMM_BIGMIN") && atol(getenv("UNIVST_GEMM_BIGMIN"))) ? atol(getenv("UNIVST_GEMM_BIGMIN")) : 150};   // measured cross-over (tools/bench_gemm_mid.py)
    return e;
}
// ragged: a plain linear (no GEGLU, no LayerNorm fold / statistics, no split-K) may end in a partly filled column tile (N % 8 == 0):
// the MM-DiT widths of the SD3 path (1536, 4608, 6144 = 4.8 / 14.4 / 19.2 tiles) lose 4 % of the MFMA work to padding and still run
// 1.4x faster than on the 128 x 128 tile.  Weight rows >= N come from the zero page; both epilogues skip columns >= N.
// (round 5: ragged_min — convolutions of widths 256 / 512 (the temporal VAE: a last column tile 80 % / 60 % full) also take the 256x320 tile: 900+ TFLOP/s of
// useful work against ~500 on the 128 x 128 tile; UNIVST_CONV_RAGGED=0 switches that off)
static bool big_shape_ok(int N, int K, long x_elems, bool ragged = false, int ragged_min = 640) {      // x_elems: extent of the activation operand in elements (32-bit DMA offsets)
    return !big_env().nobig && (N % 320 == 0 || (ragged && N % 8 == 0 && N >= ragged_min)) && (long)N * K < (1L << 31) && x_elems < (1L << 31);
}

// Does a plain linear [M, K] (row stride ldx, 0 = K) x [N, K]^T take the direct (no split-K) 256x320 path whose epilogue can fold a
// LayerNorm / emit row statistics?  Same conditions as the launcher (which still re-checks and fails loudly on a mismatch), incl.
// the LDS epilogue switch; pointer alignment is the caller's business (the UNet arena is 256-byte aligned).
bool uv_linear_takes_big_direct(long M, int N, int K, long ldx) {
    if (!big_shape_ok(N, K, M * (ldx ? ldx : (long)K)) || big_env().epi == 0) return false;
    const long n256 = ((M + 255) / 256) * (N / 320), n192 = ((M + 191) / 192) * (N / 320);
    return n256 >= big_env().bigmin && n192 >= big_env().bigmin;      // whichever tile height the launcher picks
}

int uv_launch_geglu_xres_permute(const half_t* in, half_t* out, int rows, int cols, hipStream_t stream) {
    UV_REQUIRE(rows % 256 == 0 && cols >= 1, "geglu_xres_permute: %d rows must be a multiple of 256", rows);
    hipLaunchKernelGGL(geglu_xres_permute_kernel, dim3((unsigned)(((long)rows * cols + 255) / 256)), dim3(256), 0, stream, in, out, rows, cols);
    UV_LAUNCH_CHECK();
    return UV_OK;
}
// M = 0: the shape alone (weight preparation).  With M: a block walks ALL column tiles of its 128 rows (74 us at N = 2 560), so the grid
// has to fill whole rounds of the CUs: at least two rounds, the last one >= 80 % full — one rank of an 8-GPU job (192 blocks) and a 1.5-round
// grid stay on the 256x320 tile (emulated rank: 1.34 vs 1.41 ms for the class).  UNIVST_GEGLU_XRES=0: never, =2: whenever the shape fits.
bool uv_geglu_xres_ok(int N, int K, long M) {
    static const int env = getenv("UNIVST_GEGLU_XRES") ? atoi(getenv("UNIVST_GEGLU_XRES")) : 1;
    if (env == 0 || K != XR_K || N % XR_BN != 0) return false;
    if (M <= 0 || env == 2) return true;
    const long blocks = (M + XR_BM - 1) / XR_BM, ncu = uv_num_cus(), rounds = (blocks + ncu - 1) / ncu;
    return blocks >= 2 * ncu && blocks * 10 >= rounds * ncu * 8;
}

// ONE copy of how the 128-wide path tiles a problem and whether it would split K (used by the launcher and by the fold predicates).
struct SmallPlan {
    bool nf5, small_m;
    int bn, nt, splits;
};
static SmallPlan small_plan(long M, int N, int K, bool geglu, int mode, bool want_stats) {
    SmallPlan sp;
    // NF = 5 (160-column tiles) for widths that 160 divides and 128 does not — and for every producer of row statistics: the
    // block's 160 columns are then exactly one statistics slot
    sp.nf5 = !geglu && (N % 160 == 0) && ((N % 128 != 0) || want_stats);
    sp.bn = sp.nf5 ? 160 : 128;
    sp.nt = (int)(((M + BM_DEFAULT - 1) / BM_DEFAULT) * ((N + sp.bn - 1) / sp.bn));
    // small-M problems (deepest UNet level: 3072 rows): 64-row tiles double the block count so the chip is filled
    // UNIVST_GEMM_SMALLM (A/B aid): 0 = never, 1 = whenever the 128-row tiles are < 2 per CU, 2 (default) = convs only when
    // split-K cannot supply the parallelism instead (short K); linears always (measured: tools/bench_gemm_mid.py)
    static const int smallm_mode = getenv("UNIVST_GEMM_SMALLM") ? atoi(getenv("UNIVST_GEMM_SMALLM")) : 2;
    sp.small_m = !sp.nf5 && sp.nt < 2 * uv_num_cus() && M > 64 && smallm_mode != 0 &&
                 (smallm_mode == 1 || mode == 0 || geglu || (K + 63) / 64 < 16);
    if (sp.small_m) sp.nt = (int)(((M + 63) / 64) * ((N + sp.bn - 1) / sp.bn));
    // split-K when the tiles alone leave most CUs idle and K is long (deep levels; every level of a frame shard)
    sp.splits = 1;
    static const int splitk = getenv("UNIVST_GEMM_SPLITK") ? atoi(getenv("UNIVST_GEMM_SPLITK")) : 1;
    if (splitk && !geglu && N % 4 == 0 && sp.nt < 384) {
        const int nk = (K + 63) / 64;
        const int s = uv_pick_splits(sp.nt, nk, 2L * uv_num_cus(), 4, 16, 1.0, (double)M * N * 4.0);     // two resident blocks per CU
        if (s >= 2) sp.splits = s;
    }
    return sp;
}

// May a LayerNorm be folded around a plain linear [M, K] x [N, K]^T?  As the PRODUCER of the normalised tensor it has to leave the row
// statistics of its output (one slot per 160 columns), as the CONSUMER it applies them in its epilogue.  Both exist in the direct
// 256x320 kernel and, since round 4, in the 128-wide kernel when that runs the problem without split-K (the epilogue of a split
// problem lives in the reduction kernel).  The GEGLU consumer exists in the 256x320 kernel only.
bool uv_linear_fold_producer_ok(long M, int N, int K) {
    if (uv_linear_takes_big_direct(M, N, K)) return true;
    static const int env = getenv("UNIVST_LN_FOLD_SMALL") ? atoi(getenv("UNIVST_LN_FOLD_SMALL")) : 1;
    if (!env || N % 160 != 0 || K % 8 != 0) return false;
    if (big_shape_ok(N, K, M * (long)K)) {      // the 256x320 path with split-K would take it (long reductions): no epilogue there
        const long n256 = ((M + 255) / 256) * (N / 320);
        if (n256 >= 8 && K >= 128 * 64) return false;
    }
    return small_plan(M, N, K, false, 0, true).splits == 1;
}
bool uv_linear_fold_consumer_ok(long M, int N, int K, bool geglu) {
    if (uv_linear_takes_big_direct(M, N, K)) return true;
    static const int env = getenv("UNIVST_LN_FOLD_SMALL") ? atoi(getenv("UNIVST_LN_FOLD_SMALL")) : 1;
    if (!env || geglu || N % 4 != 0 || K % 160 != 0) return false;
    if (big_shape_ok(N, K, M * (long)K)) {
        const long n256 = ((M + 255) / 256) * (N / 320);
        if (n256 >= 8 && K >= 128 * 64) return false;
    }
    return small_plan(M, N, K, false, 0, false).splits == 1;
}

// algorithmic (compulsory) bytes of a launch for the roofline leg: A operand + weights + output (+ the residual it reads).  A 3x3 conv reads its INPUT
// tensor once — the im2col-expanded operand M x 9Cin (what the matrix pipe consumes, served from LDS / L2 re-reads) is reported beside it as `aux`.
static inline void uv_gemm_bytes(const GemmParams& p, int mode, double* compulsory, double* expanded) {
    const double out = (double)p.M * (p.geglu ? p.N / 2 : p.N), w = (double)p.N * p.K, res = p.R ? (double)p.M * p.N : 0.0;
    const double a_exp = (double)p.M * p.K;
    double a = a_exp;
    if (mode == 1 && p.Ho > 0 && p.Wo > 0) a = (double)(p.M / ((long)p.Ho * p.Wo)) * p.Hs * p.Ws * (p.C1 + p.C2);
    *compulsory = 2.0 * (a + w + out + res);
    *expanded = 2.0 * (a_exp + w + out + res);
}


int uv_launch_gemm(const GemmParams& p0, int mode, hipStream_t stream) {
    GemmParams p = p0;
    if (mode == 1 && p.tapw == 0) {       // kernel geometry defaults: 3x3 with padding 1, 1x1 without
        p.tapw = p.taps == 9 ? 3 : 1;
        p.pady = p.padx = p.taps == 9 ? 1 : 0;
    }
    UV_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
    if (p.geglu == 2) {                  // weights in the X-resident kernel's row order (geglu_xres_kernel)
        auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
        UV_REQUIRE(mode == 0 && p.K == XR_K && p.N % XR_BN == 0 && p.ldx % 8 == 0 && p.ldy % 8 == 0 && al16(p.X) && al16(p.W) && al16(p.Y) && al16(p.bias) &&
                   !p.R && !p.rowbias && !p.bias2 && !p.stats_out && !p.act && !p.gate && !p.gn_out && (long)p.M * p.ldx < (1L << 31) &&
                   (!p.ln_stats || (p.ln_wsum && p.ln_bias && p.ln_slots > 0 && !p.bias)),
                   "geglu (X-resident order): needs K = 320, N %% 256 == 0, 16-byte aligned rows, no residual / second bias (M=%d N=%d K=%d)", p.M, p.N, p.K);
        uv_prof_begin(UV_CLS_GEMM_BIG, 2.0 * p.M * (double)p.N * p.K, 2.0 * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * (p.N / 2)), stream,
                      p.ln_stats ? "geglu_xres_kernel<2>" : "geglu_xres_kernel<0>");
        const dim3 grid((p.M + XR_BM - 1) / XR_BM);
        if (p.ln_stats) hipLaunchKernelGGL((geglu_xres_kernel<2>), grid, dim3(512), 0, stream, p);
        else hipLaunchKernelGGL((geglu_xres_kernel<0>), grid, dim3(512), 0, stream, p);
        uv_prof_end(stream);
        UV_LAUNCH_CHECK();
        return UV_OK;
    }
    const bool lnf = p.ln_stats || p.stats_out;
    UV_REQUIRE(p.K % 8 == 0, "gemm: K=%d must be a multiple of 8", p.K);
    UV_REQUIRE(!(p.act || p.gate) || (!p.geglu && !lnf && mode == 0), "gemm: the activation / gate epilogue is for plain linears (no GEGLU, LayerNorm fold, conv)");
    UV_REQUIRE(!p.gate || (p.rows_per_gate >= 1 && p.ld_gate % 8 == 0 && p.N % 8 == 0 && (reinterpret_cast<uintptr_t>(p.gate) & 15) == 0),
               "gemm: gate rows must be 16-byte aligned (ld_gate %% 8 == 0, N %% 8 == 0)");
    if (mode == 0) {
        UV_REQUIRE(p.ldx % 8 == 0, "gemm: ldx=%ld must be a multiple of 8", p.ldx);
    } else {
        UV_REQUIRE(p.C1 % 8 == 0 && p.C2 % 8 == 0, "conv: channel counts must be multiples of 8 (C1=%d C2=%d)", p.C1, p.C2);
        UV_REQUIRE((p.taps == 1 && p.tapw == 1) || (p.taps == 9 && p.tapw == 3) || (p.taps == 3 && p.tapw == 1), "conv: taps=%d with %d per row (1x1, 3x3 or the 3x1 frame conv)", p.taps, p.tapw);
        UV_REQUIRE(p.K == p.taps * (p.C1 + p.C2), "conv: K=%d != taps*(C1+C2)", p.K);
        const bool sym3 = p.taps == 9 && p.pady == 1 && p.padx == 1;
        UV_REQUIRE(!p.korder || (sym3 && p.C1 % 64 == 0 && p.C2 % 64 == 0), "conv: tap-inner k order needs the 3x3 / padding-1 kernel and 64-channel slabs");
        UV_REQUIRE(!p.W32 || sym3, "conv: the LDS-patch weight copy is for the 3x3 / padding-1 kernel");
    }
    {   // large-M path: 256x320 tiles when they tile N exactly and fill the chip (>= 2 blocks per CU)
        const int nobig = big_env().nobig;
        // tile height: 256 rows, or 192 when that fills whole rounds of the CUs better (49152 x 640 is 384 tiles of 256 = 1.5
        // rounds but 512 tiles of 192 = 2 exact ones; 12288 x 1280 is 192 vs 256 tiles)
        static const int bm_env = getenv("UNIVST_GEMM_BM") ? atoi(getenv("UNIVST_GEMM_BM")) : 0;     // A/B aid: force 256 / 192
        const long ncu = uv_num_cus();
        static const int conv_ragged = getenv("UNIVST_CONV_RAGGED") ? atoi(getenv("UNIVST_CONV_RAGGED")) : 1;
        // a ragged last column tile: plain linears (N >= 640) and, since round 5, plain convs whose width fills a whole number of 64-column wave halves
        // and at least 80 % of one tile (256, 512, 576, ...: not the UNet's widths, which are multiples of 320)
        const bool conv_rag = mode == 1 && conv_ragged && !p.gn_out && p.N % 64 == 0 && p.N >= 256 && (p.N % 320 == 0 || p.N % 320 >= 192);
        const bool ragged = (mode == 0 && !p.geglu && !lnf) || conv_rag;
        const int ntn_c = (p.N + 319) / 320;
        const long n256 = (long)((p.M + 255) / 256) * ntn_c, n192 = (long)((p.M + 191) / 192) * ntn_c;
        // per-row cost of the 192-row tile relative to the 256-row one, measured at equal round counts: convs 0.96-1.0, linears 1.02-1.07
        // (round 5: the conv factor was 0.99, which at the 64x64 level — 768 tiles of 256 rows = 3 full rounds against 1 024 of 192 = 4 — picked 192
        // by a hair; measured on the LDS-patch kernel the 256-row tile is 2.5 - 4 % faster there (0.357 / 0.936 / 0.585 ms against 0.366 / 0.973 /
        // 0.609 for 320->320, 960->320, 640->320 at 64 x 64), while the 32x32 and 16x16 levels keep 192: fewer rounds.  UNIVST_CONV_192_COST: A/B aid)
        static const double conv192 = getenv("UNIVST_CONV_192_COST") ? atof(getenv("UNIVST_CONV_192_COST")) : 1.03;
        const double c256 = (double)((n256 + ncu - 1) / ncu) * 256.0, c192 = (double)((n192 + ncu - 1) / ncu) * 192.0 * (mode == 1 ? conv192 : 1.05);
        // (the MM-DiT epilogue instantiation fits the 256-VGPR budget only with the 192-row tile: 223 registers; 256 rows spill 48)
        const bool mmdit_epi = mode == 0 && (p.act || p.gate);
        bool use192 = mmdit_epi ? true : (bm_env ? bm_env == 192 : (c192 < c256 && n192 >= 150));
        if (p.w_rows_per_set) {           // weight sets per row range: a tile must lie inside one set
            UV_REQUIRE(mode == 0 && p.bias32 && !p.geglu && !p.ln_stats && !mmdit_epi && p.M % p.w_rows_per_set == 0 && (p.w_rows_per_set % 256 == 0 || p.w_rows_per_set % 192 == 0) &&
                       (long)(p.M / p.w_rows_per_set) * p.N * p.K + (long)p.N * p.K < (1L << 31),
                       "linear: weight sets need a plain linear, an fp32 bias per set and %d rows per set that a 256- or 192-row tile divides", p.w_rows_per_set);
            if (p.w_rows_per_set % (use192 ? 192 : 256) != 0) use192 = !use192;
        }
        const long nblk = use192 ? n192 : n256;
        const long xmax = (mode == 0) ? (long)p.M * p.ldx : (long)p.M * (p.C1 > p.C2 ? p.C1 : p.C2) * 4;
        const long bigmin = big_env().bigmin;
        // few tiles but a long reduction (the 8x8-level convs; most convs of a frame shard): the big tile with split-K
        int bsplits = 1;
        static const int splitk_big = getenv("UNIVST_GEMM_SPLITK") ? atoi(getenv("UNIVST_GEMM_SPLITK")) : 1;
        if (splitk_big && !nobig && !p.geglu && p.N % 320 == 0 && nblk < bigmin && nblk >= 8 && p.K >= 128 * 64) {   // fp32 partials cost ~35 us: long reductions only
            const int nk = (p.K + 63) / 64;
            int sp = uv_pick_splits(nblk, nk, uv_num_cus(), 24, 8, 2.2, (double)p.M * p.N * 4.0);
            while (sp >= 2 && (size_t)sp * p.M * p.N * sizeof(float) > UV_SPLITK_WS_BYTES) --sp;     // what the partial workspace holds
            if (sp >= 2 && nblk * sp >= 128) bsplits = sp;
        }
        static const int patch_env = getenv("UNIVST_CONV_PATCH") ? atoi(getenv("UNIVST_CONV_PATCH")) : 1;
        const bool use_patch = patch_env && mode == 1 && (nblk >= bigmin || bsplits > 1) && uv_conv_patch_eligible(p, use192 ? 192 : 256);
        UV_REQUIRE(p.W || use_patch, "conv: only the [Cin/32][9][32] weight copy was given but the problem is not eligible for the LDS-patch kernel "
                   "(3x3, stride 1, whole image rows per 256/192-row tile, >= 150 tiles or a reduction long enough for split-K)");
        if (big_shape_ok(p.N, p.K, xmax, ragged, conv_rag ? 256 : 640) && (nblk >= bigmin || bsplits > 1)) {
            char sym[48];
            if (use_patch) snprintf(sym, sizeof sym, "conv_patch_kernel<%d>", use192 ? 3 : 4);
            else snprintf(sym, sizeof sym, "gemm_big_kernel<%d,%d,%d>", mode, use192 ? 3 : 4, mode == 0 ? (p.ln_stats ? 2 : (p.stats_out ? 1 : (mmdit_epi ? 3 : 0))) : 0);
            double by, byx;
            uv_gemm_bytes(p, mode, &by, &byx);
            uv_prof_begin(mode == 0 ? UV_CLS_GEMM_BIG : (use_patch ? UV_CLS_CONV_PATCH : UV_CLS_CONV_BIG), 2.0 * p.M * (double)p.N * p.K, by, stream, sym, byx);
            // row-contiguous epilogue through LDS needs 16-byte aligned rows everywhere it touches; it pays for the plain
            // and residual epilogues (-12..19 % at K=320) but not for GEGLU, whose stores are half as many (UNIVST_GEMM_EPI=2 forces it)
            const int epi = big_env().epi;
            auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
            GemmParams q = p;
            {   // row-group x column-block tile order for wide outputs (see gemm_big_kernel); UNIVST_GEMM_TILEORDER=0: column-fastest
                static const int to_env = getenv("UNIVST_GEMM_TILEORDER") ? atoi(getenv("UNIVST_GEMM_TILEORDER")) : 1;
                const int ntn = ntn_c;
                q.tile_gn = 0;
                if (to_env && mode == 0 && ntn > 4 && (long)p.N * p.K * 2 > (3L << 20)) {      // W larger than ~3 MB: it cannot stay in L2 as a whole
                    q.tile_gn = ntn % 2 == 0 ? 2 : 0;       // 8 x 2 measured best of 8x4 / 16x2 / 4x8 / 16x4 / 32x4 (all within 2 %); 10 x 3 was a loss
                    q.tile_gm = 8;
                    if (to_env > 1) { q.tile_gn = to_env % 100; q.tile_gm = to_env / 100; if (ntn % q.tile_gn) q.tile_gn = 0; }    // A/B: gm*100 + gn
                }
            }
            // next tile's DMA: activation rows before the first k-half's MFMAs, weight rows before the second (1, default; +2..8 % on
            // the linears: the LDS-DMA writes at 64 B/clk and competes with the fragment reads) or all at once (0)
            static const int issue_mode = getenv("UNIVST_GEMM_ISSUE") ? atoi(getenv("UNIVST_GEMM_ISSUE")) : 1;
            q.issue_mode = issue_mode;
            q.epi_lds = epi && (!p.geglu || epi == 2) && p.ldy % 8 == 0 && al16(p.Y) && al16(p.bias) && al16(p.bias2) && al16(p.rowbias) && p.ldrb % 8 == 0 &&
                        (!p.R || (p.ldr % 8 == 0 && al16(p.R)));
            // GEGLU (round 4): register math + fp16 slab + row-contiguous 16-byte stores (epi_lds = 3); UNIVST_GEMM_EPI=4 keeps the 8-byte register stores (A/B)
            const bool geglu_slab = p.geglu && mode == 0 && epi != 0 && epi != 2 && epi != 4 && p.ldy % 8 == 0 && al16(p.Y) && al16(p.bias) && p.N % 16 == 0;
            if (geglu_slab) q.epi_lds = 3;
            // the fp32 bias (GroupNorm folded into the linear) exists in the row epilogue through LDS only: a launch that would land on the register
            // epilogue (misaligned Y / R, UNIVST_GEMM_EPI=0) or on split-K must not drop it silently
            UV_REQUIRE(!p.bias32 || (q.epi_lds == 1 && bsplits <= 1), "linear: the fp32 bias needs the LDS row epilogue of the direct path (16-byte aligned Y / R / bias rows, no split-K)");
            // (Measured and rejected on this tile, DESIGN.md §4: a 32-wide-k 4-stage DMA ring with counted vmcnt (-10 %), the same
            // with two wave groups staggered by half a k tile + s_setprio (-0..18 %), and a five-phase / two-barriers-per-phase
            // schedule with in-place restaging two k tiles ahead (the guide's 8-phase template on this shape: -3 % linears,
            // -20 % convs).)
            bool own_ws = false;
            q.splits = 1;
            if (bsplits > 1) {
                const int nk = (p.K + 63) / 64;
                q.ktps = (nk + bsplits - 1) / bsplits;
                if (use_patch) q.ktps = (q.ktps + 8) / 9 * 9;      // a split starts at a 64-channel slab pair (9 k tiles)
                q.splits = (nk + q.ktps - 1) / q.ktps;
                const size_t need = (size_t)q.splits * p.M * p.N * sizeof(float);
                if (!(q.partial && q.partial_bytes >= need)) {
                    UV_HIP(hipMallocAsync((void**)&q.partial, need, stream));
                    own_ws = true;
                }
            }
            UV_REQUIRE(!p.w_rows_per_set || (q.epi_lds == 1 && q.splits == 1), "linear: weight sets run on the direct 256x320 path with the LDS epilogue only");
            if (p.gn_out) {       // GroupNorm statistics from this epilogue: LDS epilogue, whole 160-column halves of 10 / 20 / 40-channel groups, no split-K
                const bool ok = q.epi_lds == 1 && q.splits == 1 && !p.geglu && !p.stats_out && !p.act && !p.gate && p.N % 320 == 0 && p.M % 16 == 0 &&
                                p.gn_G > 0 && p.gn_gw * p.gn_G == p.N && (p.gn_gw == 10 || p.gn_gw == 20 || p.gn_gw == 40);
                if (!ok) q.gn_out = nullptr;
                else if (p.gn_emitted) *p.gn_emitted = 1;
            }
            const dim3 bgrid((unsigned)(nblk * q.splits));
            if (use_patch) {
                static const int wissue = getenv("UNIVST_CONV_PATCH_WISSUE") ? atoi(getenv("UNIVST_CONV_PATCH_WISSUE")) : 1;
                q.issue_mode = wissue ? 1 : 0;
            }
            if (use_patch) {          // 3x3 / stride 1 on whole image rows: input patch in LDS, k order [Cin/32][9][32] (p.W32)
                if (use192) hipLaunchKernelGGL((conv_patch_kernel<3>), bgrid, dim3(512), 0, stream, q);
                else hipLaunchKernelGGL((conv_patch_kernel<4>), bgrid, dim3(512), 0, stream, q);
                if (q.splits > 1) {
                    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(((long)p.M * (p.N / 4) + 255) / 256)), dim3(256), 0, stream, q);
                    if (own_ws) UV_HIP(hipFreeAsync(q.partial, stream));
                }
                uv_prof_end(stream);
                UV_LAUNCH_CHECK();
                return UV_OK;
            }
            // (Round 4, measured and removed: a PERSISTENT form of this tile for the GEGLU projections — one block per CU walking its tiles, the next
            // tile's first k tile DMA'd into the free buffer before the epilogue — 22.8 vs 22.6 us per K = 320 tile, -2.5 % at K >= 640
            // (256 VGPRs + 8 spilled dwords): block relaunch and prologue latency are not what the 13 us of fixed cost are made of.  The
            // kernel is kept as tools/probes/gemm_big_persist_kernel.inc for the record.)
            if (lnf) {                // LayerNorm folded into this linear / row statistics emitted for the next one
                if (p.geglu && !geglu_slab) q.epi_lds = 0;
                UV_REQUIRE(mode == 0 && q.splits == 1 && (q.epi_lds || !p.stats_out) && (!p.ln_stats || (p.ln_wsum && p.ln_bias && p.ln_slots > 0)) &&
                           (!p.stats_out || (!p.geglu && p.N % 160 == 0)), "linear: LayerNorm fold on a problem the direct 256x320 path does not take");
                UV_REQUIRE(!(p.ln_stats && p.stats_out), "linear: a LayerNorm-folded linear cannot also emit row statistics");
                if (p.ln_stats) {
                    if (use192) hipLaunchKernelGGL((gemm_big_kernel<0, 3, 2>), bgrid, dim3(512), 0, stream, q);
                    else hipLaunchKernelGGL((gemm_big_kernel<0, 4, 2>), bgrid, dim3(512), 0, stream, q);
                } else {
                    if (use192) hipLaunchKernelGGL((gemm_big_kernel<0, 3, 1>), bgrid, dim3(512), 0, stream, q);
                    else hipLaunchKernelGGL((gemm_big_kernel<0, 4, 1>), bgrid, dim3(512), 0, stream, q);
                }
            } else if (mmdit_epi) {                            // MM-DiT epilogue: GELU(tanh) / gate (.) + residual
                hipLaunchKernelGGL((gemm_big_kernel<0, 3, 3>), bgrid, dim3(512), 0, stream, q);
            } else if (use192) {
                if (mode == 0) hipLaunchKernelGGL((gemm_big_kernel<0, 3>), bgrid, dim3(512), 0, stream, q);
                else hipLaunchKernelGGL((gemm_big_kernel<1, 3>), bgrid, dim3(512), 0, stream, q);
            } else if (mode == 0) hipLaunchKernelGGL((gemm_big_kernel<0, 4>), bgrid, dim3(512), 0, stream, q);
            else hipLaunchKernelGGL((gemm_big_kernel<1, 4>), bgrid, dim3(512), 0, stream, q);
            if (q.splits > 1) {
                hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(((long)p.M * (p.N / 4) + 255) / 256)), dim3(256), 0, stream, q);
                if (own_ws) UV_HIP(hipFreeAsync(q.partial, stream));
            }
            uv_prof_end(stream);
            UV_LAUNCH_CHECK();
            return UV_OK;
        }
    }
    UV_REQUIRE(!p.w_rows_per_set && !p.bias32, "linear: weight sets / an fp32 bias exist on the direct 256x320 path only (M=%d N=%d K=%d does not take it)", p.M, p.N, p.K);
    const SmallPlan plan = small_plan(p.M, p.N, p.K, p.geglu != 0, mode, p.stats_out != nullptr);
    if (lnf) {
        UV_REQUIRE(mode == 0 && !p.geglu && !p.act && !p.gate && plan.splits == 1 && !(p.ln_stats && p.stats_out) &&
                   (!p.stats_out || (plan.nf5 && p.N % 160 == 0)) &&
                   (!p.ln_stats || (p.ln_wsum && p.ln_bias && p.ln_slots > 0 && p.N % 4 == 0)),
                   "linear: LayerNorm fold requested for M=%d N=%d K=%d, which neither the direct 256x320 path nor the 128-wide path without split-K takes "
                   "(uv_linear_fold_producer_ok / uv_linear_fold_consumer_ok)", p.M, p.N, p.K);
    }
    const bool nf5 = plan.nf5, small_m = plan.small_m;
    int nt = plan.nt;
    if (p.geglu) UV_REQUIRE(p.N % 32 == 0, "geglu: N=%d must be a multiple of 32", p.N);
    GemmParams q = p;
    q.splits = 1;
    bool own_ws = false;
    if (plan.splits >= 2) {
        const int nk = (p.K + 63) / 64;
        q.ktps = (nk + plan.splits - 1) / plan.splits;
        q.splits = (nk + q.ktps - 1) / q.ktps;
        const size_t need = (size_t)q.splits * p.M * p.N * sizeof(float);
        if (q.partial && q.partial_bytes >= need) {
        } else if (need <= UV_SPLITK_WS_BYTES) {   // stand-alone operator call: stream-ordered scratch
            UV_HIP(hipMallocAsync((void**)&q.partial, need, stream));
            own_ws = true;
        } else {
            q.splits = 1;
        }
    }
    dim3 grid(nt * q.splits), block(256);
    double by, byx;
    uv_gemm_bytes(p, mode, &by, &byx);
    uv_prof_begin(mode == 0 ? UV_CLS_GEMM : UV_CLS_CONV, 2.0 * p.M * (double)p.N * p.K, by, stream, nullptr, byx);
    if (p.stats_out) {
        hipLaunchKernelGGL((gemm_kernel<5, 0, 4, 1>), grid, block, 0, stream, q);
    } else if (p.ln_stats) {
        if (small_m) hipLaunchKernelGGL((gemm_kernel<4, 0, 2, 2>), grid, block, 0, stream, q);
        else if (nf5) hipLaunchKernelGGL((gemm_kernel<5, 0, 4, 2>), grid, block, 0, stream, q);
        else hipLaunchKernelGGL((gemm_kernel<4, 0, 4, 2>), grid, block, 0, stream, q);
    } else if (small_m) {
        if (mode == 0) hipLaunchKernelGGL((gemm_kernel<4, 0, 2>), grid, block, 0, stream, q);
        else hipLaunchKernelGGL((gemm_kernel<4, 1, 2>), grid, block, 0, stream, q);
    } else if (mode == 0) {
        if (nf5) hipLaunchKernelGGL((gemm_kernel<5, 0>), grid, block, 0, stream, q);
        else hipLaunchKernelGGL((gemm_kernel<4, 0>), grid, block, 0, stream, q);
    } else {
        if (nf5) hipLaunchKernelGGL((gemm_kernel<5, 1>), grid, block, 0, stream, q);
        else hipLaunchKernelGGL((gemm_kernel<4, 1>), grid, block, 0, stream, q);
    }
    if (q.splits > 1) {
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(((long)p.M * (p.N / 4) + 255) / 256)), dim3(256), 0, stream, q);
        if (own_ws) UV_HIP(hipFreeAsync(q.partial, stream));
    }
    uv_prof_end(stream);
    UV_LAUNCH_CHECK();
    return UV_OK;
}

int uv_launch_linear_small(const half_t* x, const half_t* W, const half_t* b, half_t* y, int M, int N, int K,
                           int silu_in, hipStream_t stream) {
    UV_REQUIRE(M >= 1 && M <= 8, "linear_small: M=%d must be in 1..8", M);
    UV_REQUIRE(K % 8 == 0, "linear_small: K=%d must be a multiple of 8", K);
    hipLaunchKernelGGL(linear_small_kernel, dim3((N + 3) / 4), dim3(256), 0, stream, x, W, b, y, M, N, K, silu_in);
    UV_LAUNCH_CHECK();
    return UV_OK;
}
