// AdaIN-guided attention-feature shift (the PnP injection) and latent AdaIN — HBM-bound kernels with
// wavefront-shuffle reductions.
//
// attention shift on the fused QKV buffer [3*F*N rows, 3C] (branch 0 = content, 1 = style, 2 = stylised):
//   Q2 = gamma * (alpha*Q0 + (1-alpha)*Q2)
//   K2 = beta * (LN_C(K2) * sigma_s + mu_s) + (1-beta) * K1          (same for V)
// where mu_s / sigma_s are the per-(frame, channel) mean / UNBIASED std of the style branch over the
// N tokens, and LN_C is the affine-free LayerNorm over the channel axis per (frame, token) — the
// reference's F.instance_norm layout quirk, reproduced on purpose.
// Replaces pnp_utils.py:44-57 + attention_adain :114-125, and latent_adain :128-139.
#include "common.h"
#include "kernels.h"

namespace {

// per-(frame, column) mean and unbiased std over N rows.  block = 8 column-chunks x 32 row phases.
// Sums are taken about a per-thread PIVOT (the first value the thread sees) and the 32 row phases are merged Chan-style in double
// ((n, mean, M2) triples): a raw one-pass sum(x), sum(x^2) in fp32 cancels when |mean| >> std, and style K / V columns of real
// checkpoints are not zero-mean (same reason as the GroupNorm pivot in norm.hip; tests: test_attention_adain_shift_large_mean).
__global__ __launch_bounds__(256) void colstats_kernel(const half_t* __restrict__ x, long ld, int N, int ncols,
                                                       float* __restrict__ mean, float* __restrict__ stdv) {
    __shared__ float sm[3][32][64];
    const int tc = threadIdx.x & 7, tr = threadIdx.x >> 3;
    const int f = blockIdx.y, c0 = blockIdx.x * 64 + tc * 8;
    float s[8], q[8], pv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = q[e] = pv[e] = 0.f;
    if (c0 < ncols && tr < N) {
        const half_t* base = x + (long)f * N * ld + c0;
        {
            h8 v = *reinterpret_cast<const h8*>(base + (long)tr * ld);
#pragma unroll
            for (int e = 0; e < 8; ++e) pv[e] = (float)v[e];
        }
        // eight rows per iteration, their loads issued together: with one 16-byte load in flight per thread the 160 blocks of the 64x64 level ran at
        // 1 TB/s (83 us for 84 MB; round 5)
        int r = tr + 32;
        for (; r + 224 < N; r += 256) {
            h8 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const h8*>(base + (long)(r + 32 * u) * ld);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float t = (float)v[u][e] - pv[e];
                    s[e] += t;
                    q[e] += t * t;
                }
        }
        for (; r < N; r += 32) {
            h8 v = *reinterpret_cast<const h8*>(base + (long)r * ld);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = (float)v[e] - pv[e];
                s[e] += t;
                q[e] += t * t;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sm[0][tr][tc * 8 + e] = s[e];
        sm[1][tr][tc * 8 + e] = q[e];
        sm[2][tr][tc * 8 + e] = pv[e];
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c = blockIdx.x * 64 + threadIdx.x;
        if (c < ncols) {
            double n = 0.0, mu = 0.0, m2 = 0.0;
            for (int t = 0; t < 32 && t < N; ++t) {
                const double nt = (double)((N - t + 31) / 32);
                const double st = sm[0][t][threadIdx.x], qt = sm[1][t][threadIdx.x];
                const double mt = (double)sm[2][t][threadIdx.x] + st / nt, m2t = qt - st * st / nt;
                const double dl = mt - mu, ntot = n + nt;
                m2 += m2t + dl * dl * n * nt / ntot;
                mu += dl * nt / ntot;
                n = ntot;
            }
            double var = m2 / (N > 1 ? (N - 1) : 1);
            if (var < 0.0) var = 0.0;
            mean[(long)f * ncols + c] = (float)mu;
            stdv[(long)f * ncols + c] = (float)sqrt(var);
        }
    }
}

template <int MAXCH>
__global__ __launch_bounds__(256) void adain_shift_kernel(half_t* __restrict__ qkv, long ld, int F, int N, int C,
                                                          const float* __restrict__ mean, const float* __restrict__ stdv,
                                                          float alpha, float beta, float gamma) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long FN = (long)F * N;
    if (r >= FN) return;
    const int f = (int)(r / N);
    const int nch = C / 8;
    half_t* row0 = qkv + r * ld;
    half_t* row1 = qkv + (FN + r) * ld;
    half_t* row2 = qkv + (2 * FN + r) * ld;
    // Q
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
            h8 a = *reinterpret_cast<const h8*>(row0 + ch * 8);
            h8 b = *reinterpret_cast<const h8*>(row2 + ch * 8);
            h8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (half_t)(gamma * (alpha * (float)a[e] + (1.f - alpha) * (float)b[e]));
            *reinterpret_cast<h8*>(row2 + ch * 8) = o;
        }
    }
    // K (t=0) and V (t=1)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int off = (1 + t) * C;
        h8 x[MAXCH];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) {
                x[i] = *reinterpret_cast<const h8*>(row2 + off + ch * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) s += (float)x[i][e];
            }
        }
        const float mu = wave_sum(s) / C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float dlt = (float)x[i][e] - mu;
                    q += dlt * dlt;
                }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / C + 1e-5f);
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) {
                h8 sty = *reinterpret_cast<const h8*>(row1 + off + ch * 8);
                const float* mp = mean + (long)f * 2 * C + t * C + ch * 8;
                const float* sp = stdv + (long)f * 2 * C + t * C + ch * 8;
                h8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float ad = ((float)x[i][e] - mu) * rstd * sp[e] + mp[e];
                    o[e] = (half_t)(beta * ad + (1.f - beta) * (float)sty[e]);
                }
                *reinterpret_cast<h8*>(row2 + off + ch * 8) = o;
            }
        }
    }
}

// latent_adain on [1,Cl,F,h,w] fp16: one block per channel.
__global__ __launch_bounds__(1024) void latent_adain_kernel(const half_t* __restrict__ cnt, const half_t* __restrict__ sty,
                                                            half_t* __restrict__ out, int F, int HW) {
    __shared__ float red[2][32];
    __shared__ float st[2];
    const int c = blockIdx.x;
    const long base = (long)c * F * HW;
    const int n = F * HW;
    auto block_sum2 = [&](float a, float b, float& ra, float& rb) {
        a = wave_sum(a);
        b = wave_sum(b);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) {
            red[0][threadIdx.x >> 6] = a;
            red[1][threadIdx.x >> 6] = b;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double x = 0, y = 0;
            for (int i = 0; i < (int)(blockDim.x >> 6); ++i) {
                x += red[0][i];
                y += red[1][i];
            }
            st[0] = (float)x;
            st[1] = (float)y;
        }
        __syncthreads();
        ra = st[0];
        rb = st[1];
    };
    // content: biased var over (F,h,w); two passes for accuracy
    float s = 0.f, dummy = 0.f, tot, t2;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += (float)cnt[base + i];
    block_sum2(s, dummy, tot, t2);
    const float cmu = tot / n;
    float q = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        float d = (float)cnt[base + i] - cmu;
        q += d * d;
    }
    block_sum2(q, dummy, tot, t2);
    const float crstd = rsqrtf(tot / n + 1e-5f);
    for (int f = 0; f < F; ++f) {
        const half_t* sp = sty + base + (long)f * HW;
        float a = 0.f;
        for (int i = threadIdx.x; i < HW; i += blockDim.x) a += (float)sp[i];
        block_sum2(a, dummy, tot, t2);
        const float smu = tot / HW;
        float b = 0.f;
        for (int i = threadIdx.x; i < HW; i += blockDim.x) {
            float d = (float)sp[i] - smu;
            b += d * d;
        }
        block_sum2(b, dummy, tot, t2);
        const float sstd = sqrtf(tot / (HW > 1 ? HW - 1 : 1));
        for (int i = threadIdx.x; i < HW; i += blockDim.x) {
            long o = base + (long)f * HW + i;
            out[o] = (half_t)(((float)cnt[o] - cmu) * crstd * sstd + smu);
        }
    }
}

// frame-shard halves of latent_adain: (1) per-channel {sum, sumsq} of the local content frames, (2) apply with
// the globally reduced statistics.  One block per channel.
__global__ __launch_bounds__(1024) void latent_adain_stats_kernel(const half_t* __restrict__ cnt, float* __restrict__ st, int n) {
    __shared__ float red[2][16];
    const int c = blockIdx.x;
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        float v = (float)cnt[(long)c * n + i];
        a += v;
        b += v * v;
    }
    a = wave_sum(a);
    b = wave_sum(b);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = a;
        red[1][threadIdx.x >> 6] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double x = 0, y = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) {
            x += red[0][i];
            y += red[1][i];
        }
        st[c * 2] = (float)x;
        st[c * 2 + 1] = (float)y;
    }
}
__global__ __launch_bounds__(1024) void latent_adain_apply_kernel(const half_t* __restrict__ cnt, const half_t* __restrict__ sty,
                                                                  const float* __restrict__ st, double n_total,
                                                                  half_t* __restrict__ out, int F, int HW) {
    __shared__ float red[2][16];
    __shared__ float bc[2];
    const int c = blockIdx.x;
    const long base = (long)c * F * HW;
    const double mu = st[c * 2] / n_total;
    double var = st[c * 2 + 1] / n_total - mu * mu;
    if (var < 0) var = 0;
    const float cmu = (float)mu, crstd = (float)(1.0 / sqrt(var + 1e-5));
    for (int f = 0; f < F; ++f) {
        const half_t* sp = sty + base + (long)f * HW;
        float a = 0.f, b = 0.f;
        for (int i = threadIdx.x; i < HW; i += blockDim.x) {
            float v = (float)sp[i];
            a += v;
            b += v * v;
        }
        a = wave_sum(a);
        b = wave_sum(b);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) {
            red[0][threadIdx.x >> 6] = a;
            red[1][threadIdx.x >> 6] = b;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double x = 0, y = 0;
            for (int i = 0; i < (int)(blockDim.x >> 6); ++i) {
                x += red[0][i];
                y += red[1][i];
            }
            double m = x / HW, v = (y - x * m) / (HW > 1 ? HW - 1 : 1);
            bc[0] = (float)m;
            bc[1] = (float)sqrt(v > 0 ? v : 0);
        }
        __syncthreads();
        const float smu = bc[0], sstd = bc[1];
        for (int i = threadIdx.x; i < HW; i += blockDim.x) {
            long o = base + (long)f * HW + i;
            out[o] = (half_t)(((float)cnt[o] - cmu) * crstd * sstd + smu);
        }
    }
}

}  // namespace

int uv_launch_latent_adain_stats(const half_t* cnt, float* st, int Cl, int F, int HW, hipStream_t stream) {
    hipLaunchKernelGGL(latent_adain_stats_kernel, dim3(Cl), dim3(1024), 0, stream, cnt, st, F * HW);
    UV_LAUNCH_CHECK();
    return UV_OK;
}
int uv_launch_latent_adain_apply(const half_t* cnt, const half_t* sty, const float* st, long n_total, half_t* out, int Cl, int F, int HW,
                                 hipStream_t stream) {
    hipLaunchKernelGGL(latent_adain_apply_kernel, dim3(Cl), dim3(1024), 0, stream, cnt, sty, st, (double)n_total, out, F, HW);
    UV_LAUNCH_CHECK();
    return UV_OK;
}

int uv_launch_colstats(const half_t* x, long ld, int F, int N, int ncols, float* mean, float* stdv, hipStream_t stream) {
    UV_REQUIRE(ncols % 8 == 0 && ld % 8 == 0, "colstats: ncols/ld must be multiples of 8");
    hipLaunchKernelGGL(colstats_kernel, dim3((ncols + 63) / 64, F), dim3(256), 0, stream, x, ld, N, ncols, mean, stdv);
    UV_LAUNCH_CHECK();
    return UV_OK;
}

int uv_launch_adain_shift(half_t* qkv, long ld, int F, int N, int C, float* mean, float* stdv, float alpha, float beta,
                          float gamma, hipStream_t stream) {
    UV_REQUIRE(C % 8 == 0 && C <= 2048 && ld % 8 == 0, "adain_shift: C=%d ld=%ld unsupported", C, ld);
    // style statistics: K1|V1 columns [C,3C) of the style branch rows [F*N, 2*F*N)
    uv_prof_begin(UV_CLS_ADAIN, 0.0, 2.0 * (double)F * N * C * 11.0, stream);   // 6 reads + 3 writes + 2 stat reads of [F,N,C] fp16
    int rc = uv_launch_colstats(qkv + (long)F * N * ld + C, ld, F, N, 2 * C, mean, stdv, stream);
    if (rc) return rc;
    dim3 grid((unsigned)(((long)F * N + 3) / 4)), block(256);
    const int nch = (C / 8 + 63) / 64;
    if (nch <= 1) hipLaunchKernelGGL((adain_shift_kernel<1>), grid, block, 0, stream, qkv, ld, F, N, C, mean, stdv, alpha, beta, gamma);
    else if (nch == 2) hipLaunchKernelGGL((adain_shift_kernel<2>), grid, block, 0, stream, qkv, ld, F, N, C, mean, stdv, alpha, beta, gamma);
    else hipLaunchKernelGGL((adain_shift_kernel<4>), grid, block, 0, stream, qkv, ld, F, N, C, mean, stdv, alpha, beta, gamma);
    uv_prof_end(stream);
    UV_LAUNCH_CHECK();
    return UV_OK;
}

int uv_launch_latent_adain(const half_t* cnt, const half_t* sty, half_t* out, int Cl, int F, int HW, hipStream_t stream) {
    hipLaunchKernelGGL(latent_adain_kernel, dim3(Cl), dim3(1024), 0, stream, cnt, sty, out, F, HW);
    UV_LAUNCH_CHECK();
    return UV_OK;
}
