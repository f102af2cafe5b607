// First vertical slice of the SD3 / SD3.5 rectified-flow path (BASELINE config 5): the pieces the REFERENCE owns —
// its two joint-attention processors and the rectified-flow inversion updates — on the native kernels.  The MM-DiT around them
// (diffusers' SD3Transformer2DModel) is third-party and not built; nothing here claims a backbone.
//
//   uv_sd3_joint_attention   CrossFrameProcessor (video_diffusion_sd3/pnp_utils.py:17-131) and AttentionShiftProcessor (:143-271,
//                            under the documented fixed reading thresh2 == eta2, oracle/sd3_ref.py): q/k/v + added q/k/v
//                            projections (MFMA GEMMs), per-head RMSNorm of q / k, the AdaIN-guided shift of the stylised branch,
//                            cross-frame K/V ['first', f-1, f] read BY POINTER plus the text tokens as one extra key segment of
//                            its own length (AttnParams::kx), output projections.
//   rms_heads_kernel         diffusers RMSNorm over the head dim (qk_norm = "rms_norm"), in place on a [rows, heads*d] slice
//   sd3 AdaIN shift          F.instance_norm on [B, heads, N, d] normalises over (N, d) JOINTLY per (frame, head) (biased variance,
//                            eps 1e-5), re-coloured with the style branch's per-(frame, head, channel) mean / unbiased std over N
//   adaln_modulate_kernel    LayerNorm(no affine) * (1 + scale[b]) + shift[b]  (AdaLayerNormZero / AdaLayerNormContinuous)
//   axpbypcz                 the three-term updates of rf_inversion / rf_solver (inversion_tools/flow_inversion.py:123-264)
#include <math.h>

#include "common.h"
#include "kernels.h"
#include "unet.h"
#include "../../include/univst.h"

namespace {

// one wave per (row, head): x <- x * rsqrt(mean(x^2) + eps) * w      (d <= 256, multiple of 2)
__global__ __launch_bounds__(256) void rms_heads_kernel(half_t* __restrict__ x, long ld, long rows, int heads, int d,
                                                        const half_t* __restrict__ w, float eps, float scl) {
    const long unit = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (unit >= rows * heads) return;
    const long r = unit / heads;
    const int h = (int)(unit - r * heads);
    half_t* p = x + r * ld + (long)h * d;
    float v[4];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = lane + 64 * i;
        v[i] = e < d ? (float)p[e] : 0.f;
        ss += v[i] * v[i];
    }
    const float r_ = rsqrtf(wave_sum(ss) / d + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = lane + 64 * i;
        if (e < d) p[e] = (half_t)(v[i] * r_ * (float)w[e] * scl);
    }
}

// the same for head widths of 8 * 2^k (32, 64, 128): LPH = d / 8 lanes share a head, 16 bytes per lane, the reduction is
// log2(LPH) shuffles; one pass serves BOTH the q and the k columns of a fused q | k | v row (column blocks c0 and c1, own weights;
// c1 < 0: one block).  HBM-bound: 2 x 2 x rows x heads x d bytes per call.
template <int LPH>
__global__ __launch_bounds__(256) void rms_heads_vec_kernel(half_t* __restrict__ x, long ld, long rows, int heads, long c0, long c1,
                                                            const half_t* __restrict__ w0, const half_t* __restrict__ w1, float eps, float scl0) {
    constexpr int D = LPH * 8;
    const long per_row = (long)heads * LPH;                       // 16-byte pieces per row and column block
    const long nblk = c1 >= 0 ? 2 : 1;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * per_row * nblk) return;                       // (whole heads sit inside a wave: 64 % LPH == 0, per_row % LPH == 0)
    const long r = i / (per_row * nblk);
    const long q = i - r * per_row * nblk;
    const int blk = (int)(q / per_row);
    const long piece = q - blk * per_row;
    const int e0 = (int)(piece % LPH) * 8;
    half_t* ptr = x + r * ld + (blk ? c1 : c0) + piece * 8;
    const h8 v = *reinterpret_cast<const h8*>(ptr);
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) ss = fmaf((float)v[e], (float)v[e], ss);
#pragma unroll
    for (int o = 1; o < LPH; o <<= 1) ss += __shfl_xor(ss, o, 64);
    const float r_ = rsqrtf(ss / D + eps) * (blk ? 1.f : scl0);      // scl0: the attention's scale * log2(e) rides on the q block (one rounding)
    const h8 wv = *reinterpret_cast<const h8*>((blk ? w1 : w0) + e0);
    h8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)v[e] * r_ * (float)wv[e]);
    *reinterpret_cast<h8*>(ptr) = o;
}

// launcher: q and k of a fused buffer in one pass when the head width allows (w1 / c1 optional)
static int launch_rms_pair(half_t* x, long ld, long rows, int heads, int d, long c0, const half_t* w0, long c1, const half_t* w1, float eps,
                           hipStream_t s, float scl0 = 1.f) {
    const bool two = w1 != nullptr;
    const bool vec = (d == 32 || d == 64 || d == 128) && ld % 8 == 0 && c0 % 8 == 0 && (!two || c1 % 8 == 0) &&
                     ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w0) | reinterpret_cast<uintptr_t>(w1)) & 15) == 0;
    if (vec) {
        const long n = rows * heads * (d / 8) * (two ? 2 : 1);
        const dim3 grid((unsigned)((n + 255) / 256));
        const long cc1 = two ? c1 : -1;
        if (d == 32) hipLaunchKernelGGL(rms_heads_vec_kernel<4>, grid, dim3(256), 0, s, x, ld, rows, heads, c0, cc1, w0, w1, eps, scl0);
        else if (d == 64) hipLaunchKernelGGL(rms_heads_vec_kernel<8>, grid, dim3(256), 0, s, x, ld, rows, heads, c0, cc1, w0, w1, eps, scl0);
        else hipLaunchKernelGGL(rms_heads_vec_kernel<16>, grid, dim3(256), 0, s, x, ld, rows, heads, c0, cc1, w0, w1, eps, scl0);
    } else {
        hipLaunchKernelGGL(rms_heads_kernel, dim3((unsigned)((rows * heads + 3) / 4)), dim3(256), 0, s, x + c0, ld, rows, heads, d, w0, eps, scl0);
        if (two) hipLaunchKernelGGL(rms_heads_kernel, dim3((unsigned)((rows * heads + 3) / 4)), dim3(256), 0, s, x + c1, ld, rows, heads, d, w1, eps, 1.f);
    }
    UV_LAUNCH_CHECK();
    return UV_OK;
}

// per (frame, t in {K, V}, head): (mu, rstd) of the stylised branch over (N, d) jointly, from its per-column statistics
// (colstats: mean_c, unbiased std_c over the N rows): sum_n x^2 = (N-1) std_c^2 + N mean_c^2
__global__ void sd3_group_stats_kernel(const float* __restrict__ mean, const float* __restrict__ stdv, int F, int N, int C, int heads,
                                       float* __restrict__ mu, float* __restrict__ rstd) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;       // (f, t, h)
    if (i >= F * 2 * heads) return;
    const int d = C / heads;
    const float* m = mean + (long)i * d;                       // columns of [F][2C] are (t, h, e)-major: i*d is exactly (f, t, h, 0)
    const float* s = stdv + (long)i * d;
    double a = 0.0, b = 0.0;
    for (int e = 0; e < d; ++e) {
        const double me = m[e], se = s[e];
        a += me;
        b += (double)(N - 1) * se * se + (double)N * me * me;
    }
    const double mean_g = a / d;
    double var = b / ((double)N * d) - mean_g * mean_g;
    if (var < 0.0) var = 0.0;
    mu[i] = (float)mean_g;
    rstd[i] = (float)(1.0 / sqrt(var + 1e-5));
}

// the shift itself, one wave per token row of the stylised branch (rows [2*F*N, 3*F*N) of the fused [3*F*N, 3C] q|k|v buffer):
//   q2 <- gamma * (alpha * q0 + (1 - alpha) * q2)
//   k2 <- beta * ((k2 - mu) * rstd * sty_std + sty_mean) + (1 - beta) * k1        (same for v)
__global__ __launch_bounds__(256) void sd3_shift_kernel(half_t* __restrict__ qkv, long ld, int F, int N, int C, int heads,
                                                        const float* __restrict__ smean, const float* __restrict__ sstd,
                                                        const float* __restrict__ mu, const float* __restrict__ rstd, float alpha,
                                                        float beta, float gamma) {
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const long FN = (long)F * N;
    if (r >= FN) return;
    const int f = (int)(r / N), d = C / heads;
    half_t* row0 = qkv + r * ld;
    half_t* row1 = qkv + (FN + r) * ld;
    half_t* row2 = qkv + (2 * FN + r) * ld;
    for (int c = lane; c < C; c += 64) row2[c] = (half_t)(gamma * (alpha * (float)row0[c] + (1.f - alpha) * (float)row2[c]));
    for (int t = 0; t < 2; ++t) {
        const int off = (1 + t) * C;
        for (int c = lane; c < C; c += 64) {
            const int gi = (f * 2 + t) * heads + c / d;
            const long ci = ((long)f * 2 + t) * C + c;
            const float ad = ((float)row2[off + c] - mu[gi]) * rstd[gi] * sstd[ci] + smean[ci];
            row2[off + c] = (half_t)(beta * ad + (1.f - beta) * (float)row1[off + c]);
        }
    }
}

// src_idx [B][3] = ['first', f-1 (clipped), f] of the frame's own clip, x_idx [B] = the frame itself (pnp_utils.py:27,53-78), with
// duplicate sources merged as on the SD-v1.5 path (frame 0 of a clip reads itself three times, frame 1 reads frame 0 twice): a source
// that occurs c times is listed once with log2-weight log2(c) — softmax over duplicated keys == softmax with the key's exp weighted c.
// clip == 0: no cross-frame gather (diffusers' stock JointAttnProcessor2_0): the frame itself, once.
// Frame shard (world > 1): this rank holds frames [rank*clip, (rank+1)*clip) of every branch; the clip's first frame and the frame
// before this rank's first one are projected from the received hidden rows into the row blocks B + b (previous) and B + nbr + b (first) of branch b.
// phase (round 6, ranks > 0 of a frame shard: the two-phase attention of csrc/attention.hip): 0 = the whole key set; 1 = only the sources this rank
// HOLDS (the frame itself and, from its second local frame on, the previous one); 2 = only the halo sources (the clip's first frame and, for the rank's
// first local frame, the frame before it — merged into one source of weight 2 where they are the same frame).
__global__ void sd3_index_kernel(int B, int clip, int rank, int* __restrict__ src_idx, int* __restrict__ x_idx, int* __restrict__ cnt,
                                 float* __restrict__ logw, int phase) {
    const int bf = blockIdx.x * blockDim.x + threadIdx.x;
    if (bf >= B) return;
    x_idx[bf] = bf;
    int* si = src_idx + bf * 3;
    float* lw = logw + bf * 3;
    si[0] = si[1] = si[2] = bf;
    lw[0] = lw[1] = lw[2] = 0.f;
    if (clip == 0) {
        cnt[bf] = 1;
        return;
    }
    const int b = bf / clip, f = bf - b * clip;
    const int gf = rank * clip + f;                                   // frame index in the whole clip
    const int nbr = B / clip;
    const int first = rank == 0 ? b * clip : B + nbr + b;             // row block holding the clip's first frame (halo blocks: [previous: nbr | first: nbr], the inbox's order)
    const int prev = f > 0 ? bf - 1 : B + b;                          // (gf >= 1) row block of frame gf - 1
    if (phase == 1) {                       // rank > 0: gf >= 1
        cnt[bf] = f > 0 ? 2 : 1;
        if (f > 0) si[0] = prev;            // [prev, cur] / [cur]
        return;
    }
    if (phase == 2) {
        si[0] = first;
        if (gf == 1) {                      // 'first' and gf - 1 are the same frame
            cnt[bf] = 1;
            lw[0] = 1.f;
        } else if (f == 0) {
            cnt[bf] = 2;
            si[1] = prev;
        } else {
            cnt[bf] = 1;
        }
        return;
    }
    if (gf == 0) {
        cnt[bf] = 1;
        lw[0] = 1.5849625007211562f;          // log2(3)
    } else if (gf == 1) {
        cnt[bf] = 2;
        si[0] = first;
        lw[0] = 1.f;                          // log2(2): 'first' and gf - 1 are the same frame
    } else {
        cnt[bf] = 3;
        si[0] = first;
        si[1] = prev;
    }
}

// rows x cols8 16-byte pieces between two row-strided matrices: packs / unpacks the k | v columns of one frame
__global__ __launch_bounds__(256) void sd3_copy2d_kernel(const half_t* __restrict__ src, long ld_src, half_t* __restrict__ dst, long ld_dst,
                                                         long rows, int cols8) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * cols8) return;
    const long r = i / cols8;
    const int c = (int)(i - r * cols8) * 8;
    *reinterpret_cast<h8*>(dst + r * ld_dst + c) = *reinterpret_cast<const h8*>(src + r * ld_src + c);
}

// y = LN(x) * (1 + scale[b]) + shift[b] (and optionally y2 with a second (scale2, shift2) from the SAME normalised row:
// AdaLayerNormZeroX of the dual-attention blocks); one wave per row held in registers, C <= 4096, C % 8 == 0; scale / shift rows
// are ld_mod halfs apart (they are chunks of one [B, 6C | 9C | 2C] linear output)
__global__ __launch_bounds__(256) void adaln_modulate_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, const half_t* __restrict__ scale,
                                                             const half_t* __restrict__ shift, long ld_mod, long rows, long rows_per_batch, int C,
                                                             float eps, half_t* __restrict__ y2, const half_t* __restrict__ scale2,
                                                             const half_t* __restrict__ shift2) {
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    const long b = r / rows_per_batch;
    const half_t* xr = x + r * C;
    h8 v[8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = (lane + 64 * i) * 8;
        if (c < C) {
            v[i] = *reinterpret_cast<const h8*>(xr + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += (float)v[i][e];
        }
    }
    const float mu = wave_sum(s) / C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = (lane + 64 * i) * 8;
        if (c < C) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float dl = (float)v[i][e] - mu;
                q += dl * dl;
            }
        }
    }
    const float rs = rsqrtf(wave_sum(q) / C + eps);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = (lane + 64 * i) * 8;
        if (c < C) {
            const h8 sc = *reinterpret_cast<const h8*>(scale + b * ld_mod + c), sh = *reinterpret_cast<const h8*>(shift + b * ld_mod + c);
            h8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (half_t)(((float)v[i][e] - mu) * rs * (1.f + (float)sc[e]) + (float)sh[e]);
            *reinterpret_cast<h8*>(y + r * C + c) = o;
            if (y2) {
                const h8 sc2 = *reinterpret_cast<const h8*>(scale2 + b * ld_mod + c), sh2 = *reinterpret_cast<const h8*>(shift2 + b * ld_mod + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (half_t)(((float)v[i][e] - mu) * rs * (1.f + (float)sc2[e]) + (float)sh2[e]);
                *reinterpret_cast<h8*>(y2 + r * C + c) = o;
            }
        }
    }
}

// out[r, c] = x[r, c] + gate[b(r), c] * y[r, c]  (the gated residuals of the MM-DiT block; gate rows ld_gate halfs apart); C % 8 == 0
__global__ __launch_bounds__(256) void gate_residual_kernel(const half_t* __restrict__ x, const half_t* __restrict__ gate, long ld_gate,
                                                            const half_t* __restrict__ y, half_t* __restrict__ out, long rows, long rows_per_batch,
                                                            int C) {
    const int c8 = C / 8;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * c8) return;
    const long r = i / c8;
    const int c = (int)(i - r * c8) * 8;
    const long b = r / rows_per_batch;
    const h8 xv = *reinterpret_cast<const h8*>(x + r * C + c), yv = *reinterpret_cast<const h8*>(y + r * C + c);
    const h8 gv = *reinterpret_cast<const h8*>(gate + b * ld_gate + c);
    h8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)fmaf((float)gv[e], (float)yv[e], (float)xv[e]);
    *reinterpret_cast<h8*>(out + r * C + c) = o;
}

// act 0: SiLU  x * sigmoid(x);  act 1: GELU(tanh)  0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))   (torch gelu approximate='tanh')
template <int ACT>
__global__ __launch_bounds__(256) void act_kernel(const half_t* __restrict__ x, half_t* __restrict__ out, long n8) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    const h8 v = reinterpret_cast<const h8*>(x)[i];
    h8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float t = (float)v[e];
        if (ACT == 0) o[e] = (half_t)(t / (1.f + __expf(-t)));
        else {
            const float u = 0.7978845608028654f * fmaf(0.044715f * t * t, t, t);
            o[e] = (half_t)(0.5f * t * (1.f + tanhf(u)));
        }
    }
    reinterpret_cast<h8*>(out)[i] = o;
}

// diffusers get_timestep_embedding(t, dim, flip_sin_to_cos, downscale_freq_shift, scale 1, max_period 10000) in fp32, stored fp16:
// freq_i = exp(-ln(max_period) * i / (dim/2 - shift)); [sin | cos] halves, swapped when flip
__global__ void timestep_embed_kernel(const float* __restrict__ t, half_t* __restrict__ out, int B, int dim, int flip, float shift, float max_period) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int half_dim = dim / 2;
    if (i >= B * half_dim) return;
    const int b = i / half_dim, j = i - b * half_dim;
    const float freq = expf(-logf(max_period) * (float)j / ((float)half_dim - shift));
    const float a = t[b] * freq;
    const float sn = sinf(a), cs = cosf(a);
    out[(long)b * dim + j] = (half_t)(flip ? cs : sn);
    out[(long)b * dim + half_dim + j] = (half_t)(flip ? sn : cs);
}

// PatchEmbed's Conv2d(Cc, D, kernel p, stride p) as a linear: rows[(b, i, j)][c*p*p + u*p + v] = lat[b, c, i*p + u, j*p + v]
// (k index order = the flattened conv weight [D][Cc][p][p]); FORWARD: latent -> rows, else rows -> latent (unpatchify: the row holds
// [u][v][c] as diffusers' reshape (h, w, p, p, c) -> einsum nhwpqc->nchpwq does)
template <bool FORWARD>
__global__ __launch_bounds__(256) void patch_kernel(half_t* __restrict__ lat, half_t* __restrict__ rowsb, int B, int Cc, int H, int W, int P) {
    const long n = (long)B * Cc * H * W;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int w = (int)(i % W), h = (int)(i / W % H), c = (int)(i / ((long)W * H) % Cc);
    const long b = i / ((long)W * H * Cc);
    const int pi = h / P, u = h - pi * P, pj = w / P, v = w - pj * P;
    const long row = (b * (H / P) + pi) * (W / P) + pj;
    const int K = Cc * P * P;
    if (FORWARD) rowsb[row * K + (c * P + u) * P + v] = lat[i];
    else lat[i] = rowsb[row * K + (u * P + v) * Cc + c];
}

__global__ void axpbypcz_kernel(const half_t* __restrict__ x, const half_t* __restrict__ y, const half_t* __restrict__ z,
                                half_t* __restrict__ out, float a, float b, float c, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (half_t)(a * (float)x[i] + b * (float)y[i] + c * (float)z[i]);
}

int linear(const half_t* X, long ldx, long M, int K, const half_t* W, const half_t* b, int N, half_t* Y, long ldy, hipStream_t s,
           const half_t* res = nullptr, const half_t* gate = nullptr, long ld_gate = 0, int rows_per_gate = 1) {
    GemmParams g;
    g.X = X; g.ldx = ldx; g.M = (int)M; g.K = K; g.N = N; g.W = W; g.bias = b; g.Y = Y; g.ldy = ldy;
    g.R = res; g.ldr = N; g.gate = gate; g.ld_gate = ld_gate; g.rows_per_gate = rows_per_gate;      // Y = res + gate[b] (.) (X W^T + b)
    return uv_launch_gemm(g, 0, s);
}

}  // namespace

#define RUN(x)                \
    do {                      \
        int _rc = (x);        \
        if (_rc) return _rc;  \
    } while (0)

extern "C" {

int univst_rmsnorm_heads(void* x, int64_t ld, int64_t rows, int heads, int d, const void* weight, float eps, void* stream) {
    UV_REQUIRE(x && weight && rows >= 1 && heads >= 1 && d >= 1 && d <= 256, "rmsnorm_heads: bad argument (head_dim <= 256)");
    return launch_rms_pair((half_t*)x, ld, rows, heads, d, 0, (const half_t*)weight, 0, nullptr, eps, (hipStream_t)stream);
}

int univst_adaln_modulate(const void* x, void* y, const void* scale, const void* shift, int64_t ld_mod, int64_t rows, int64_t rows_per_batch,
                          int C, float eps, void* y2, const void* scale2, const void* shift2, void* stream) {
    UV_REQUIRE(x && y && scale && shift && rows >= 1 && rows_per_batch >= 1 && C >= 8 && C % 8 == 0 && C <= 4096 && ld_mod % 8 == 0 && ld_mod >= C,
               "adaln_modulate: bad argument (C=%d must be a multiple of 8, <= 4096; ld_mod=%ld)", C, (long)ld_mod);
    UV_REQUIRE(!y2 || (scale2 && shift2), "adaln_modulate: y2 needs scale2 and shift2");
    hipLaunchKernelGGL(adaln_modulate_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, (half_t*)y,
                       (const half_t*)scale, (const half_t*)shift, (long)ld_mod, (long)rows, (long)rows_per_batch, C, eps, (half_t*)y2,
                       (const half_t*)scale2, (const half_t*)shift2);
    UV_LAUNCH_CHECK();
    return UV_OK;
}

int univst_gate_residual(const void* x, const void* gate, int64_t ld_gate, const void* y, void* out, int64_t rows, int64_t rows_per_batch, int C,
                         void* stream) {
    UV_REQUIRE(x && gate && y && out && rows >= 1 && rows_per_batch >= 1 && C >= 8 && C % 8 == 0 && ld_gate % 8 == 0, "gate_residual: bad argument");
    const long n = rows * (C / 8);
    hipLaunchKernelGGL(gate_residual_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const half_t*)x,
                       (const half_t*)gate, (long)ld_gate, (const half_t*)y, (half_t*)out, (long)rows, (long)rows_per_batch, C);
    UV_LAUNCH_CHECK();
    return UV_OK;
}

int univst_activation(const void* x, void* out, int64_t n, int act, void* stream) {
    UV_REQUIRE(x && out && n >= 8 && n % 8 == 0 && (act == UNIVST_ACT_SILU || act == UNIVST_ACT_GELU_TANH), "activation: bad argument");
    const long n8 = n / 8;
    if (act == UNIVST_ACT_SILU)
        hipLaunchKernelGGL(act_kernel<0>, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, (half_t*)out, n8);
    else
        hipLaunchKernelGGL(act_kernel<1>, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, (half_t*)out, n8);
    UV_LAUNCH_CHECK();
    return UV_OK;
}

int univst_timestep_embedding(const float* t, void* out, int B, int dim, int flip_sin_to_cos, float downscale_freq_shift, float max_period,
                              void* stream) {
    UV_REQUIRE(t && out && B >= 1 && dim >= 2 && dim % 2 == 0 && max_period > 1.f, "timestep_embedding: bad argument");
    const int n = B * (dim / 2);
    hipLaunchKernelGGL(timestep_embed_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, (hipStream_t)stream, t, (half_t*)out, B, dim,
                       flip_sin_to_cos, downscale_freq_shift, max_period);
    UV_LAUNCH_CHECK();
    return UV_OK;
}

int univst_sd3_patchify(const void* latents, void* rows, int B, int C, int H, int W, int patch, void* stream) {
    UV_REQUIRE(latents && rows && B >= 1 && C >= 1 && patch >= 1 && H % patch == 0 && W % patch == 0, "sd3_patchify: bad argument");
    const long n = (long)B * C * H * W;
    hipLaunchKernelGGL(patch_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (half_t*)latents, (half_t*)rows, B, C,
                       H, W, patch);
    UV_LAUNCH_CHECK();
    return UV_OK;
}

int univst_sd3_unpatchify(const void* rows, void* latents, int B, int C, int H, int W, int patch, void* stream) {
    UV_REQUIRE(latents && rows && B >= 1 && C >= 1 && patch >= 1 && H % patch == 0 && W % patch == 0, "sd3_unpatchify: bad argument");
    const long n = (long)B * C * H * W;
    hipLaunchKernelGGL(patch_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (half_t*)latents, (half_t*)rows, B, C,
                       H, W, patch);
    UV_LAUNCH_CHECK();
    return UV_OK;
}

int univst_axpbypcz(const void* x, const void* y, const void* z, void* out, float a, float b, float c, int64_t n, void* stream) {
    UV_REQUIRE(x && y && z && out && n >= 1, "axpbypcz: bad argument");
    hipLaunchKernelGGL(axpbypcz_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, (const half_t*)y,
                       (const half_t*)z, (half_t*)out, a, b, c, n);
    UV_LAUNCH_CHECK();
    return UV_OK;
}

// in place on the fused [3*F*N, 3C] q | k | v buffer (row stride ld; branch 0 content, 1 style, 2 stylised); ws: 4*F*2C + 2*F*2*heads floats
int univst_sd3_adain_shift(void* qkv, int64_t ld, int F, int N, int C, int heads, float alpha, float beta, float gamma, void* ws, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    UV_REQUIRE(qkv && ws && F >= 1 && N >= 2 && heads >= 1 && C % heads == 0 && C % 8 == 0, "sd3_adain_shift: bad argument");
    half_t* q = (half_t*)qkv;
    float *smean = (float*)ws, *sstd = smean + (long)F * 2 * C, *cmean = sstd + (long)F * 2 * C, *cstd = cmean + (long)F * 2 * C;
    float *mu = cstd + (long)F * 2 * C, *rstd = mu + (long)F * 2 * heads;
    const long FN = (long)F * N;
    RUN(uv_launch_colstats(q + FN * ld + C, ld, F, N, 2 * C, smean, sstd, s));           // style branch K | V: per (frame, channel) over N
    RUN(uv_launch_colstats(q + 2 * FN * ld + C, ld, F, N, 2 * C, cmean, cstd, s));       // stylised branch K | V
    hipLaunchKernelGGL(sd3_group_stats_kernel, dim3((unsigned)((F * 2 * heads + 127) / 128)), dim3(128), 0, s, cmean, cstd, F, N, C, heads, mu, rstd);
    hipLaunchKernelGGL(sd3_shift_kernel, dim3((unsigned)((FN + 3) / 4)), dim3(256), 0, s, q, (long)ld, F, N, C, heads, smean, sstd, mu, rstd, alpha,
                       beta, gamma);
    UV_LAUNCH_CHECK();
    return UV_OK;
}

int univst_sd3_joint_attention(const univst_sd3_attn_weights* w, const void* hidden, const void* enc, int B, int N, int Nt, int Cin, int heads,
                               int head_dim, int clip_length, int shift, float beta, float rms_eps, void* out_img,
                               void* out_txt, const univst_sd3_gated_residual* gr, univst_comm* comm, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    UV_REQUIRE(w && hidden && out_img && B >= 1 && N >= 1 && heads >= 1, "sd3_joint_attention: null / empty argument");
    UV_REQUIRE(w->to_q && w->to_k && w->to_v && w->to_out, "sd3_joint_attention: to_q / to_k / to_v / to_out weights are required");
    UV_REQUIRE(clip_length >= 0 && (clip_length == 0 || B % clip_length == 0), "sd3_joint_attention: batch %d is not a whole number of %d-frame clips",
               B, clip_length);
    UV_REQUIRE(!shift || B % 3 == 0, "sd3_joint_attention: the attention shift needs the three-branch batch");
    UV_REQUIRE(!enc || (out_txt && Nt >= 1 && w->add_q && w->add_k && w->add_v), "sd3_joint_attention: text tokens need add_{q,k,v}_proj and out_txt");
    UV_REQUIRE(!gr || (gr->res_img && gr->gate_img && (!enc || !w->to_add_out || (gr->res_txt && gr->gate_txt))),
               "sd3_joint_attention: a gated residual needs residual and gate pointers for every projected stream");
    const int C = heads * head_dim;
    UV_REQUIRE(Cin % 8 == 0 && C % 8 == 0, "sd3_joint_attention: widths must be multiples of 8");
    const int world = comm ? uv_comm_world(comm) : 1, rank = comm ? uv_comm_rank(comm) : 0;
    const bool sharded = world > 1;
    UV_REQUIRE(!sharded || clip_length >= 1, "sd3_joint_attention: a frame shard needs the cross-frame processors (clip_length = frames per rank)");
    const int nbr = clip_length ? B / clip_length : 0;                 // branches (clips) in the batch
    const long rows_i = (long)B * N, rows_t = enc ? (long)B * Nt : 0;
    const long rows_x = sharded ? (long)2 * nbr * N : 0;               // halo row blocks behind the local frames: (first, previous) per branch
    // one stream-ordered scratch block: qkv_img [rows_i, 3C] | qkv_txt [rows_t, 3C] | o_img [rows_i, C] | o_txt [rows_t, C] | stats | index tables
    const int Fb = B / 3;                                    // frames per branch (shift only)
    const size_t n_half = (size_t)(rows_i + rows_t) * 4 * C + (size_t)rows_x * 3 * C;
    const int Fst = Fb > 1 ? Fb : 1;
    const size_t n_stat = shift ? (size_t)4 * Fst * 2 * C + (size_t)2 * Fst * 2 * heads : 0;
    char* ws = nullptr;
    const bool two_phase = sharded && rank > 0;              // round 6: the keys this rank holds first, the halo frames continue from the softmax state
    const size_t n_state = two_phase ? (size_t)(rows_i + rows_t) * heads * 2 : 0;
    const size_t bytes = n_half * sizeof(half_t) + (n_stat + n_state) * sizeof(float) + (size_t)B * 8 * sizeof(int) + 1024;      // src_idx[3B] x_idx[B] cnt[B] logw[3B]
    UV_HIP(hipMallocAsync((void**)&ws, bytes, s));
    half_t* qkv_i = (half_t*)ws;
    half_t* qkv_t = qkv_i + (rows_i + rows_x) * 3 * C;
    half_t* o_i = qkv_t + rows_t * 3 * C;
    half_t* o_t = o_i + rows_i * C;
    float* st = (float*)(((uintptr_t)(o_t + rows_t * C) + 255) & ~(uintptr_t)255);
    float* state_i = st + n_stat;
    float* state_t = state_i + (two_phase ? (size_t)rows_i * heads * 2 : 0);
    int* tab = (int*)(st + n_stat + n_state);
    auto H = [](const void* p) { return (const half_t*)p; };
    int rc = UV_OK;
    auto body = [&]() -> int {
        const half_t* x = H(hidden);
        // ONE q | k | v projection per stream when the three weight matrices (and biases) are consecutive in memory — the host mirror
        // keeps them so (univst_amd/_native.py: a [3C, Cin] copy per attention module): the activation rows are read once instead of three
        // times and the 256 x 320 tile runs 14.4 column tiles instead of 3 x 4.8 (round 4)
        auto fused3 = [&](const void* q, const void* k, const void* v, const void* qb, const void* kb, const void* vb) {
            const half_t *q_ = H(q), *k_ = H(k), *v_ = H(v);
            const bool wf = k_ == q_ + (long)C * Cin && v_ == k_ + (long)C * Cin;
            const bool bf = (!qb && !kb && !vb) || (qb && H(kb) == H(qb) + C && H(vb) == H(kb) + C);
            return wf && bf;
        };
        // ---- frame shard (round 6): the exchange in two packs of different kinds, both on the communicator's forked stream.
        //   previous frame (this rank's last frame of every branch -> rank + 1): its HIDDEN rows — this op's input, the adaLN-modulated tokens every
        //     projection reads — [branch][N][Cin], half a K | V pack, posted before anything else of the layer; the RECEIVER projects them to K | V,
        //     applies the k RMSNorm and (inside the window) the per-frame AdaIN shift itself (pnp_utils.py:183-194);
        //   the clip's first frame (rank 0 -> every rank): its finished K | V [branch][N][2C], posted by rank 0 once it has projected, normalised and
        //     shifted its own rows — every rank needs the same rows, so nobody projects them twice, and each link from rank 0 carries one such pack.
        // Rank 1's one link from rank 0 carries both (hidden rows first); with 64 GB/s links and the config-5 shape both land before the local phase of the
        // attention ends on every rank.  Workspace: 64 KiB of all-reduce scratch, then send | first | inbox[parity][previous, first], slots of the larger pack.
        long o_prev = 0, o_rfirst = 0, o_first = 0;
        char* ws_c = sharded ? uv_comm_ws(comm) : nullptr;
        hipStream_t xs = s;
        const long bytes_h = (long)nbr * N * Cin * (long)sizeof(half_t), bytes_kv = (long)nbr * N * 2 * C * (long)sizeof(half_t);
        const bool emu = sharded && uv_comm_emulated(comm);
        if (sharded) {
            const long slot = (((bytes_h > bytes_kv ? bytes_h : bytes_kv)) + 255) & ~255L;
            UV_REQUIRE(65536 + 6 * slot <= uv_comm_ws_bytes(comm), "sd3_joint_attention: the communicator's workspace (%ld bytes) is smaller than 64 KiB + "
                       "6 packs of %ld bytes", uv_comm_ws_bytes(comm), slot);
            const long o_send = 65536;
            o_first = 65536 + slot;
            const unsigned par = uv_comm_kv_parity(comm);
            o_prev = 65536 + (2 + 2 * par) * slot;
            o_rfirst = o_prev + slot;
            RUN(uv_comm_kv_begin(comm));
            if (rank < world - 1) {
                const long cpn = (long)N * (Cin / 8);
                const unsigned cgrid = (unsigned)((cpn + 255) / 256);
                for (int b = 0; b < nbr; ++b)
                    hipLaunchKernelGGL(sd3_copy2d_kernel, dim3(cgrid), dim3(256), 0, s, x + ((long)(b * clip_length + clip_length - 1) * N) * Cin, (long)Cin,
                                       (half_t*)(ws_c + o_send) + (long)b * N * Cin, (long)Cin, (long)N, Cin / 8);
                UV_LAUNCH_CHECK();
            }
            if (rank < world - 1 || emu) RUN(uv_comm_fork(comm, s, &xs));
            RUN(uv_comm_kv_post_halo(comm, o_send, o_prev, bytes_h, xs));
        }
        if (fused3(w->to_q, w->to_k, w->to_v, w->to_q_bias, w->to_k_bias, w->to_v_bias)) {
            RUN(linear(x, Cin, rows_i, Cin, H(w->to_q), H(w->to_q_bias), 3 * C, qkv_i, 3 * C, s));
        } else {
            RUN(linear(x, Cin, rows_i, Cin, H(w->to_q), H(w->to_q_bias), C, qkv_i, 3 * C, s));
            RUN(linear(x, Cin, rows_i, Cin, H(w->to_k), H(w->to_k_bias), C, qkv_i + C, 3 * C, s));
            RUN(linear(x, Cin, rows_i, Cin, H(w->to_v), H(w->to_v_bias), C, qkv_i + 2 * C, 3 * C, s));
        }
        // with q / k RMSNorm (SD3.5) the attention's scale * log2(e) is applied to q inside the norm (one fp16 rounding, as the norm's
        // own output has): the attention then runs with AttnParams::q_prescaled, which the pipelined head_dim-64 kernel needs
        const float qscale = 1.4426950408889634f / sqrtf((float)head_dim);
        const bool presc = w->norm_q && w->norm_k && (!enc || (w->norm_added_q && w->norm_added_k));
        if (w->norm_q && w->norm_k) RUN(launch_rms_pair(qkv_i, 3 * C, rows_i, heads, head_dim, 0, H(w->norm_q), C, H(w->norm_k), rms_eps, s, presc ? qscale : 1.f));
        else if (w->norm_q) RUN(univst_rmsnorm_heads(qkv_i, 3 * C, rows_i, heads, head_dim, w->norm_q, rms_eps, s));
        else if (w->norm_k) RUN(univst_rmsnorm_heads(qkv_i + C, 3 * C, rows_i, heads, head_dim, w->norm_k, rms_eps, s));
        if (shift) {          // pnp_utils.py:183-194 (alpha 0.8, gamma 2.0); window test + beta come from the caller, evaluated in double
            RUN(univst_sd3_adain_shift(qkv_i, 3 * C, Fb, N, C, heads, 0.8f, beta, 2.0f, st, s));
        }
        if (sharded && (rank == 0 || emu)) {          // the clip's first frame: finished K | V rows of every branch -> every rank
            const long cpn = (long)N * (2 * C / 8);
            const unsigned cgrid = (unsigned)((cpn + 255) / 256);
            if (rank == 0) {
                for (int b = 0; b < nbr; ++b)
                    hipLaunchKernelGGL(sd3_copy2d_kernel, dim3(cgrid), dim3(256), 0, s, qkv_i + ((long)b * clip_length * N) * 3 * C + C, (long)3 * C,
                                       (half_t*)(ws_c + o_first) + (long)b * N * 2 * C, (long)2 * C, (long)N, 2 * C / 8);
                UV_LAUNCH_CHECK();
            }
            RUN(uv_comm_fork(comm, s, &xs));
            RUN(uv_comm_kv_post_first(comm, o_first, o_rfirst, bytes_kv, xs));
        }
        if (enc) {
            const half_t* e = H(enc);
            if (fused3(w->add_q, w->add_k, w->add_v, w->add_q_bias, w->add_k_bias, w->add_v_bias)) {
                RUN(linear(e, Cin, rows_t, Cin, H(w->add_q), H(w->add_q_bias), 3 * C, qkv_t, 3 * C, s));
            } else {
                RUN(linear(e, Cin, rows_t, Cin, H(w->add_q), H(w->add_q_bias), C, qkv_t, 3 * C, s));
                RUN(linear(e, Cin, rows_t, Cin, H(w->add_k), H(w->add_k_bias), C, qkv_t + C, 3 * C, s));
                RUN(linear(e, Cin, rows_t, Cin, H(w->add_v), H(w->add_v_bias), C, qkv_t + 2 * C, 3 * C, s));
            }
            if (w->norm_added_q && w->norm_added_k)
                RUN(launch_rms_pair(qkv_t, 3 * C, rows_t, heads, head_dim, 0, H(w->norm_added_q), C, H(w->norm_added_k), rms_eps, s, presc ? qscale : 1.f));
            else if (w->norm_added_q) RUN(univst_rmsnorm_heads(qkv_t, 3 * C, rows_t, heads, head_dim, w->norm_added_q, rms_eps, s));
            else if (w->norm_added_k) RUN(univst_rmsnorm_heads(qkv_t + C, 3 * C, rows_t, heads, head_dim, w->norm_added_k, rms_eps, s));
        }
        AttnParams a;
        a.k = qkv_i + C; a.v = qkv_i + 2 * C; a.ldkv = 3 * C;
        a.src_idx = tab; a.nsrc = 3; a.src_cnt = tab + 4 * B; a.src_logw = (const float*)(tab + 5 * B); a.BF = B; a.Nkv = N; a.heads = heads; a.d = head_dim;
        a.scale_log2e = qscale;
        a.q_prescaled = presc ? 1 : 0;
        auto index = [&](int phase) {
            hipLaunchKernelGGL(sd3_index_kernel, dim3((unsigned)((B + 127) / 128)), dim3(128), 0, s, B, clip_length, rank, tab, tab + 3 * B, tab + 4 * B,
                               (float*)(tab + 5 * B), phase);
        };
        auto attend = [&](int phase) -> int {                              // image queries, then text queries, over the key set of `phase`
            if (enc && phase != 2) { a.kx = qkv_t + C; a.vx = qkv_t + 2 * C; a.ldkv_x = 3 * C; a.Nkv_x = Nt; a.x_idx = tab + 3 * B; }
            else { a.kx = nullptr; a.vx = nullptr; a.x_idx = nullptr; a.Nkv_x = 0; }
            a.q = qkv_i; a.ldq = 3 * C; a.Nq = N; a.o = o_i; a.ldo = C;
            a.state_out = phase == 1 ? state_i : nullptr;
            a.state_in = phase == 2 ? state_i : nullptr;
            RUN(uv_launch_attention(a, s));                                  // image queries over [first | prev | cur] ++ text keys
            if (enc) {
                a.q = qkv_t; a.Nq = Nt; a.o = o_t;
                a.state_out = phase == 1 ? state_t : nullptr;
                a.state_in = phase == 2 ? state_t : nullptr;
                RUN(uv_launch_attention(a, s));                              // text queries over the same key set
            }
            return UV_OK;
        };
        if (!two_phase) {
            index(0);
            UV_LAUNCH_CHECK();
            RUN(attend(0));
        } else {
            index(1);
            UV_LAUNCH_CHECK();
            RUN(attend(1));                                                  // the keys this rank holds ++ the text keys, while the halo is on the wire
            RUN(uv_comm_kv_wait(comm, s));
            // the halo frames, rows behind the local ones [previous: nbr x N | first: nbr x N]: the previous frame from its hidden rows (to_k | to_v +
            // biases -> k RMSNorm -> its shift), the clip's first frame as the K | V rows rank 0 finished
            half_t* qh = qkv_i + rows_i * 3 * C;
            const long hrows = (long)nbr * N;
            const bool kvf = H(w->to_v) == H(w->to_k) + (long)C * Cin && ((!w->to_k_bias && !w->to_v_bias) || (w->to_k_bias && H(w->to_v_bias) == H(w->to_k_bias) + C));
            const half_t* hx = (const half_t*)(ws_c + o_prev);
            if (kvf) {
                RUN(linear(hx, Cin, hrows, Cin, H(w->to_k), H(w->to_k_bias), 2 * C, qh + C, 3 * C, s));
            } else {
                RUN(linear(hx, Cin, hrows, Cin, H(w->to_k), H(w->to_k_bias), C, qh + C, 3 * C, s));
                RUN(linear(hx, Cin, hrows, Cin, H(w->to_v), H(w->to_v_bias), C, qh + 2 * C, 3 * C, s));
            }
            if (w->norm_k) RUN(univst_rmsnorm_heads(qh + C, 3 * C, hrows, heads, head_dim, w->norm_k, rms_eps, s));
            if (shift)            // the previous frame of the three branches is the shift kernel's [3][F = 1][N]; its q columns are never projected and never
                                  // read (the kernel mixes them element-wise, statistics come from k / v only)
                RUN(univst_sd3_adain_shift(qh, 3 * C, 1, N, C, heads, 0.8f, beta, 2.0f, st, s));
            {
                const long cpn = hrows * (2 * C / 8);
                hipLaunchKernelGGL(sd3_copy2d_kernel, dim3((unsigned)((cpn + 255) / 256)), dim3(256), 0, s, (const half_t*)(ws_c + o_rfirst), (long)2 * C,
                                   qh + hrows * 3 * C + C, (long)3 * C, hrows, 2 * C / 8);
                UV_LAUNCH_CHECK();
            }
            index(2);
            UV_LAUNCH_CHECK();
            RUN(attend(2));                                                  // the halo frames: continues from the (m, l) state, merges
        }
        if (sharded) {
            // nobody starts the next exchange before every rank has consumed this one.  It also retires the forked stream without an event: the barrier
            // completes only after the peers consumed the packs, i.e. after this rank's multicast and raise have run
            RUN(uv_comm_barrier(comm, s));
            (void)xs;
        }
        // (gr: the block's gated residual rides in the out-projection's epilogue: out = res + gate[b] (.) to_out(o))
        RUN(linear(o_i, C, rows_i, C, H(w->to_out), H(w->to_out_bias), Cin, (half_t*)out_img, Cin, s, gr ? H(gr->res_img) : nullptr,
                   gr ? H(gr->gate_img) : nullptr, gr ? gr->ld_gate_img : 0, N));
        if (enc) {
            if (w->to_add_out)
                RUN(linear(o_t, C, rows_t, C, H(w->to_add_out), H(w->to_add_out_bias), Cin, (half_t*)out_txt, Cin, s, gr ? H(gr->res_txt) : nullptr,
                           gr ? H(gr->gate_txt) : nullptr, gr ? gr->ld_gate_txt : 0, Nt));
            else UV_HIP(hipMemcpyAsync(out_txt, o_t, (size_t)rows_t * C * sizeof(half_t), hipMemcpyDeviceToDevice, s));      // context_pre_only
        }
        return UV_OK;
    };
    rc = body();
    (void)hipFreeAsync(ws, s);
    return rc;
}

}  // extern "C"
