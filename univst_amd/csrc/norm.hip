// GroupNorm (cross-frame 5-D and per-frame 4-D statistics) + SiLU, LayerNorm, and the column-statistics
// primitive shared with the PnP AdaIN (pnp.hip).  All HBM-bound; activations are NHWC fp16 [rows, C].
//
// GroupNorm is two launches: (1) deterministic partial sums per (stat unit, row chunk, group) -> `part`
// [S, nchunk, G, 2] fp32 — this small buffer is what a multi-GPU frame shard all-reduces;
// (2) one small reduction of the chunk partials, (3) normalise + affine
// (+SiLU) + store.  The input may be the virtual channel-concat of two sources (UNet skip connection),
// groups may straddle the two; the output is one contiguous [rows, C1+C2] tensor.
// Replaces: resnet.py:338,369 + unet_3d_condition.py:439 (5-D GroupNorm eps 1e-5 + SiLU),
// attention.py:69-71,121 (per-frame GroupNorm eps 1e-6), attention.py:289-343 (LayerNorms).
#include "common.h"
#include <stdlib.h>
#include "kernels.h"

namespace {

// Threads are laid out (tr, tc): tc indexes a 16-byte column chunk (8 channels), tr a row phase.
// Each thread keeps 8 per-channel sums over its rows; block-level reduction is through LDS in a fixed
// order (deterministic).
__global__ void gn_partial_kernel(const half_t* __restrict__ s1, const half_t* __restrict__ s2, int C1, int C2,
                                  int rows_per_stat, int rows_per_chunk, int G, float* __restrict__ part) {
    // Sums are taken about a per-thread, per-channel PIVOT (the first value the thread reads): sum (x-p) and sum (x-p)^2 stay
    // of the order of the spread even when |mean| >> std (real SD checkpoints have such channel groups), where a raw
    // one-pass E[x^2] - mean^2 in fp32 cancels.  The block recombines them about zero in double, in a fixed order.
    // What this does NOT remove: the per-chunk (sum, sumsq) pair is stored — and, in a frame shard, all-reduced — as fp32, and the
    // variance is still sumsq/n - mean^2, so ONE fp32 rounding of sumsq remains: relative variance error <= (mean/std)^2 * 2^-24
    // (6e-6 at |mean| = 10 std, 5e-5 at 30 std, 6e-4 at 100 std; the per-thread accumulation error that grew with the row
    // count is what the pivot removes).  Partials as (n, mean, M2) would need a second all-reduce round in the sharded case.
    extern __shared__ float sm[];                    // [TR][C] shifted sums, [TR][C] shifted sumsq, [TR][C] pivots
    const int C = C1 + C2, TC = C / 8, TR = blockDim.x / TC;
    const int tc = threadIdx.x % TC, tr = threadIdx.x / TC;
    const int s = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int c0 = tc * 8;
    const bool second = c0 >= C1;
    const half_t* src = second ? s2 : s1;
    const int cs = second ? C2 : C1, co = second ? c0 - C1 : c0;
    const int r0 = chunk * rows_per_chunk;
    const int r1 = min(r0 + rows_per_chunk, rows_per_stat);
    float sum[8], sq[8], pv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) sum[e] = sq[e] = pv[e] = 0.f;
    if (tr < TR) {
        const long rbase = (long)s * rows_per_stat;
        int r = r0 + tr;
        if (r < r1) {
            h8 v = *reinterpret_cast<const h8*>(src + (rbase + r) * cs + co);
#pragma unroll
            for (int e = 0; e < 8; ++e) pv[e] = (float)v[e];
        }
        for (; r + 3 * TR < r1; r += 4 * TR) {
            h8 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const h8*>(src + (rbase + r + q * TR) * cs + co);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float f = (float)v[q][e] - pv[e];
                    sum[e] += f;
                    sq[e] += f * f;
                }
        }
        for (; r < r1; r += TR) {
            h8 v = *reinterpret_cast<const h8*>(src + (rbase + r) * cs + co);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = (float)v[e] - pv[e];
                sum[e] += f;
                sq[e] += f * f;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sm[tr * C + c0 + e] = sum[e];
            sm[(TR + tr) * C + c0 + e] = sq[e];
            sm[(2 * TR + tr) * C + c0 + e] = pv[e];
        }
    }
    __syncthreads();
    const int cpg = C / G;
    for (int gi = threadIdx.x; gi < G; gi += blockDim.x) {
        double a = 0.0, b = 0.0;
        for (int c = gi * cpg; c < (gi + 1) * cpg; ++c)
            for (int t = 0; t < TR; ++t) {
                const int first = r0 + t;
                const double n = first < r1 ? (double)((r1 - first + TR - 1) / TR) : 0.0;      // rows this thread-row visited
                const double p = sm[(2 * TR + t) * C + c], ds = sm[t * C + c], dq = sm[(TR + t) * C + c];
                a += n * p + ds;
                b += dq + 2.0 * p * ds + n * p * p;
            }
        float* o = part + (((long)s * nchunk + chunk) * G + gi) * 2;
        o[0] = (float)a;
        o[1] = (float)b;
    }
}

__global__ void gn_apply_kernel(const half_t* __restrict__ s1, const half_t* __restrict__ s2, int C1, int C2,
                                int rows_per_stat, int rows_per_block, int G, int nchunk, float eps, long count_rows,
                                const float* __restrict__ part, const half_t* __restrict__ gamma,
                                const half_t* __restrict__ beta, int silu, half_t* __restrict__ out) {
    extern __shared__ float sm[];                    // [G] mean, [G] rstd
    const int C = C1 + C2, TC = C / 8, TR = blockDim.x / TC;
    const int s = blockIdx.y;
    const int cpg = C / G;
    for (int gi = threadIdx.x; gi < G; gi += blockDim.x) {
        double a = 0.0, b = 0.0;
        for (int ch = 0; ch < nchunk; ++ch) {
            const float* pp = part + (((long)s * nchunk + ch) * G + gi) * 2;
            a += pp[0];
            b += pp[1];
        }
        double cnt = (double)count_rows * cpg;
        double mean = a / cnt;
        double var = b / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        sm[gi] = (float)mean;
        sm[G + gi] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const int tc = threadIdx.x % TC, tr = threadIdx.x / TC;
    if (tr >= TR) return;
    const int c0 = tc * 8;
    const bool second = c0 >= C1;
    const half_t* src = second ? s2 : s1;
    const int cs = second ? C2 : C1, co = second ? c0 - C1 : c0;
    float sc[8], sh[8];
    {
        h8 gv = *reinterpret_cast<const h8*>(gamma + c0);
        h8 bv = *reinterpret_cast<const h8*>(beta + c0);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int gi = (c0 + e) / cpg;
            float rs = sm[G + gi], mu = sm[gi];
            sc[e] = rs * (float)gv[e];
            sh[e] = (float)bv[e] - mu * sc[e];
        }
    }
    const long rbase = (long)s * rows_per_stat;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(r0 + rows_per_block, rows_per_stat);
    int r = r0 + tr;
    for (; r + 3 * TR < r1; r += 4 * TR) {          // 4 independent 16-byte loads in flight per thread
        h8 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const h8*>(src + (rbase + r + q * TR) * cs + co);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            h8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float y = (float)v[q][e] * sc[e] + sh[e];
                if (silu) y = silu_f(y);
                o[e] = (half_t)y;
            }
            *reinterpret_cast<h8*>(out + (rbase + r + q * TR) * C + c0) = o;
        }
    }
    for (; r < r1; r += TR) {
        h8 v = *reinterpret_cast<const h8*>(src + (rbase + r) * cs + co);
        h8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float y = (float)v[e] * sc[e] + sh[e];
            if (silu) y = silu_f(y);
            o[e] = (half_t)y;
        }
        *reinterpret_cast<h8*>(out + (rbase + r) * C + c0) = o;
    }
}

// ONE-LAUNCH GroupNorm for SMALL tensors (round 4; the deep UNet levels and every level of a frame shard, where the three launches of the
// streaming form above are 8 us of launch latency each for a few MB): one block per (stat unit, group).  Thread t walks rows t, t + 256, ...
// of the unit and the group's cpg channels of each row (20 - 160 B: 4-byte loads), sums about its first value (pivot) in fp32, the block
// merges the 256 (n, pivot, sum, sumsq) tuples about zero in double in a fixed order; then — MODE 0 — the same block normalises and
// stores its rows x channels (second read out of L2), or — MODE 1, cross-rank statistics — writes (sum, sumsq) for the all-reduce and the
// streaming apply kernel follows.  Two sources (virtual concat) are read per channel pair; C1 is even, so a pair never straddles them.
template <int MODE>
__global__ __launch_bounds__(256) void gn_small_kernel(const half_t* __restrict__ s1, const half_t* __restrict__ s2, int C1, int C2, int rows_per_stat,
                                                       int G, float eps, const half_t* __restrict__ gamma, const half_t* __restrict__ beta, int silu,
                                                       half_t* __restrict__ out, float* __restrict__ red) {
    const int C = C1 + C2, cpg = C / G, np = cpg >> 1;          // channel pairs per group
    const int s = blockIdx.y, gi = blockIdx.x, tid = threadIdx.x;
    const int c0 = gi * cpg;
    const long rbase = (long)s * rows_per_stat;
    __shared__ double sh[2][256];
    __shared__ float st[2];
    float pv = 0.f, s1f = 0.f, s2f = 0.f;
    int n = 0;
    bool have = false;
    for (int r = tid; r < rows_per_stat; r += 256) {
        for (int q = 0; q < np; ++q) {
            const int c = c0 + 2 * q;
            const h2 v = c < C1 ? *reinterpret_cast<const h2*>(s1 + (rbase + r) * C1 + c) : *reinterpret_cast<const h2*>(s2 + (rbase + r) * C2 + (c - C1));
            if (!have) { pv = (float)v[0]; have = true; }
            const float a = (float)v[0] - pv, b = (float)v[1] - pv;
            s1f += a + b;
            s2f = fmaf(a, a, fmaf(b, b, s2f));
            n += 2;
        }
    }
    const double dn = (double)n, dp = (double)pv;
    sh[0][tid] = dn * dp + (double)s1f;
    sh[1][tid] = (double)s2f + 2.0 * dp * (double)s1f + dn * dp * dp;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {          // fixed tree: deterministic
        if (tid < off) {
            sh[0][tid] += sh[0][tid + off];
            sh[1][tid] += sh[1][tid + off];
        }
        __syncthreads();
    }
    if (MODE == 1) {
        if (tid == 0) {
            red[((long)s * G + gi) * 2] = (float)sh[0][0];
            red[((long)s * G + gi) * 2 + 1] = (float)sh[1][0];
        }
        return;
    }
    if (tid == 0) {
        const double cnt = (double)rows_per_stat * cpg;
        const double mean = sh[0][0] / cnt;
        double var = sh[1][0] / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        st[0] = (float)mean;
        st[1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const float mu = st[0], rs = st[1];
    for (int r = tid; r < rows_per_stat; r += 256) {
        for (int q = 0; q < np; ++q) {
            const int c = c0 + 2 * q;
            const h2 v = c < C1 ? *reinterpret_cast<const h2*>(s1 + (rbase + r) * C1 + c) : *reinterpret_cast<const h2*>(s2 + (rbase + r) * C2 + (c - C1));
            const h2 gv = *reinterpret_cast<const h2*>(gamma + c), bv = *reinterpret_cast<const h2*>(beta + c);
            h2 o;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float sc = rs * (float)gv[e];
                float y = (float)v[e] * sc + ((float)bv[e] - mu * sc);
                if (silu) y = silu_f(y);
                o[e] = (half_t)y;
            }
            *reinterpret_cast<h2*>(out + (rbase + r) * C + c) = o;
        }
    }
}

// [S, nchunk, G, 2] -> [S, G, 2]: the 768-byte per-(branch, group) partial sums a frame shard all-reduces.
// One wave per output value (fixed lane-strided order + shuffle tree -> deterministic); the serial
// 128-load walk of a single thread per output cost 19 us per GroupNorm, a quarter of the whole operator.
__global__ __launch_bounds__(256) void gn_reduce_chunks_kernel(const float* __restrict__ part, float* __restrict__ red, int nchunk, int SG2,
                                                               int G2) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= SG2) return;
    const int s = i / G2, r = i - s * G2;
    double a = 0.0;                                   // (sum, sumsq about zero: cancellation-prone, keep the chunk walk in double)
    for (int c = lane; c < nchunk; c += 64) a += (double)part[((long)s * nchunk + c) * G2 + r];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
    if (lane == 0) red[i] = (float)a;
}

// Producer statistics (GemmParams::gn_out) -> [S, G, 2].  The producers leave (sum, sumsq) per 16-row fragment and per SUB-GROUP of 10 channels
// ([C/10][rows/16][2] per source tensor, sub-group major: the reduction below reads contiguous runs): 10 divides every group width of the SD-v1.5 / v2.1 UNets (10, 20, 30, 40, 60, 80), so the groups of a
// virtual channel concat (up-block resnets: 960 = 640 + 320 channels in groups of 30, ...) are whole runs of sub-groups of the two sources and the
// statistics pass over the LARGEST tensors of the step goes as well.  Deterministic (fixed strides, fixed reduction tree).
__global__ __launch_bounds__(256) void gn_reduce_sub_kernel(const float2* __restrict__ p1, const float2* __restrict__ p2, int SG1, int SG2, int nfrag,
                                                            int spg, int G, int SG, float* __restrict__ red) {
    // one BLOCK per (stat unit, group): 256 threads stride over the group's spg x nfrag partials (a single wave walking 4 096 .. 12 288 of them
    // was latency-bound: 16 - 85 us per GroupNorm), four independent loads in flight per thread, block reduction in double in a fixed order
    const int i = blockIdx.x, tid = threadIdx.x;
    const int s = i / G, gi = i - s * G;
    const long F = (long)nfrag * (SG / G);                     // fragments of the whole tensor (all stat units)
    double a = 0.0, b = 0.0;
    for (int j = 0; j < spg; ++j) {
        const int sg = gi * spg + j;
        const float2* col = (sg < SG1 ? p1 + (long)sg * F : p2 + (long)(sg - SG1) * F) + (long)s * nfrag;       // [sub-group][fragment]: contiguous
        int c = tid;
        for (; c + 768 < nfrag; c += 1024) {
            const float2 v0 = col[c], v1 = col[c + 256], v2 = col[c + 512], v3 = col[c + 768];
            a += ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
            b += ((double)v0.y + (double)v1.y) + ((double)v2.y + (double)v3.y);
        }
        for (; c < nfrag; c += 256) {
            const float2 v = col[c];
            a += (double)v.x;
            b += (double)v.y;
        }
    }
    __shared__ double sh[2][256];
    sh[0][tid] = a;
    sh[1][tid] = b;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) {
            sh[0][tid] += sh[0][tid + off];
            sh[1][tid] += sh[1][tid + off];
        }
        __syncthreads();
    }
    if (tid == 0) {
        red[(long)i * 2] = (float)sh[0][0];
        red[(long)i * 2 + 1] = (float)sh[1][0];
    }
}

// GroupNorm folded into the linear that consumes the tensor (UvGnFold): one wave per (stat unit s, output row n).  The group statistics are
// turned into (mean, rstd) exactly as gn_apply_kernel does (double), a_k = gamma_k * rstd, b_k = beta_k - mean * a_k.
__global__ __launch_bounds__(256) void gn_fold_linear_kernel(const float* __restrict__ red, const half_t* __restrict__ gamma, const half_t* __restrict__ beta,
                                                             int G, int C, long count_rows, float eps, const half_t* __restrict__ W,
                                                             const half_t* __restrict__ bias, int N, half_t* __restrict__ Wout, float* __restrict__ bias32) {
    extern __shared__ float sm[];                    // [G] mean, [G] rstd of this unit
    const int s = blockIdx.y, cpg = C / G;
    for (int gi = threadIdx.x; gi < G; gi += 256) {
        const double a = red[((long)s * G + gi) * 2], b = red[((long)s * G + gi) * 2 + 1];
        const double cnt = (double)count_rows * cpg;
        const double mean = a / cnt;
        double var = b / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        sm[gi] = (float)mean;
        sm[G + gi] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    float acc = 0.f;
    for (int k = lane; k < C; k += 64) {
        const int gi = k / cpg;
        const float w = (float)W[(long)n * C + k];
        const float ak = (float)gamma[k] * sm[G + gi];
        const float bk = (float)beta[k] - sm[gi] * ak;
        Wout[((long)s * N + n) * C + k] = (half_t)(w * ak);
        acc = fmaf(w, bk, acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) bias32[(long)s * N + n] = acc + (bias ? (float)bias[n] : 0.f);
}

// LayerNorm over the last dim, one wave per row, row kept in registers (two-pass variance).
template <int MAXCH, int RPW>
__global__ __launch_bounds__(256) void layernorm_kernel(const half_t* __restrict__ x, long ldx, half_t* __restrict__ y,
                                                        long ldy, const half_t* __restrict__ gamma,
                                                        const half_t* __restrict__ beta, int rows, int C, float eps) {
    // RPW rows per wave, all their loads issued before the first reduction: with one row per wave a wave has 640 B in flight
    // at C = 320 (24 of 64 lanes idle) and the kernel sat at 4.5 TB/s against 6.1 for a streaming add on this chip.
    const int lane = threadIdx.x & 63;
    const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= rows) return;
    const int nch = C / 8;
    h8 v[RPW][MAXCH];
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch && row0 + r < rows) v[r][i] = *reinterpret_cast<const h8*>(x + (row0 + r) * ldx + ch * 8);
        }
    h8 gv[MAXCH], bv[MAXCH];
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
            gv[i] = *reinterpret_cast<const h8*>(gamma + ch * 8);
            bv[i] = *reinterpret_cast<const h8*>(beta + ch * 8);
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const long row = row0 + r;
        if (row >= rows) break;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            if (lane + 64 * i < nch) {
#pragma unroll
                for (int e = 0; e < 8; ++e) s += (float)v[r][i][e];
            }
        }
        const float mean = wave_sum(s) / C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            if (lane + 64 * i < nch) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float d = (float)v[r][i][e] - mean;
                    q += d * d;
                }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / C + eps);
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) {
                h8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (half_t)(((float)v[r][i][e] - mean) * rstd * (float)gv[i][e] + (float)bv[i][e]);
                *reinterpret_cast<h8*>(y + row * ldy + ch * 8) = o;
            }
        }
    }
}

}  // namespace

static int gn_geometry(int C, int* block) {
    int TC = C / 8;
    int TR = 256 / TC;
    if (TR < 1) TR = 1;
    *block = TC * TR;
    return TR;
}

constexpr int GN_MAX_CHUNKS = 512;     // per stat unit; 3 x 512 blocks keep ~6 blocks per CU streaming on the big tensors
int uv_groupnorm_workspace_floats(int S, int G) { return S * (GN_MAX_CHUNKS + 1) * G * 2; }

int uv_launch_groupnorm(const half_t* s1, const half_t* s2, int C1, int C2, long rows, int rows_per_stat, int G,
                        float eps, const half_t* gamma, const half_t* beta, int silu, half_t* out, float* part,
                        hipStream_t stream, const UvGnComm* comm, const float* pre_part, const float* pre_part2, const UvGnFold* fold) {
    const int C = C1 + C2;
    UV_REQUIRE(C % 8 == 0 && C1 % 8 == 0, "groupnorm: channels must be multiples of 8 (C1=%d C2=%d)", C1, C2);
    UV_REQUIRE(C % G == 0, "groupnorm: C=%d not divisible by G=%d", C, G);
    UV_REQUIRE(C / 8 <= 1024, "groupnorm: C=%d too large", C);
    UV_REQUIRE(rows % rows_per_stat == 0, "groupnorm: rows=%ld not a multiple of rows_per_stat=%d", rows, rows_per_stat);
    const int S = (int)(rows / rows_per_stat);
    // small tensors: one launch (statistics + apply in the block that owns the (unit, group)), or statistics only when they are summed over ranks
    static const long small_bytes = getenv("UNIVST_GN_SMALL") ? atol(getenv("UNIVST_GN_SMALL")) : (4L << 20);
    const bool sharded_stats = comm && comm->world > 1;
    // producer statistics are usable when every source has them and the groups are whole runs of 10-channel sub-groups
    if (pre_part && !((!s2 || pre_part2) && (C / G) % 10 == 0 && C1 % 10 == 0 && C2 % 10 == 0 && rows_per_stat % 16 == 0)) pre_part = nullptr;
    UV_REQUIRE(!fold || (fold->W && fold->W_out && fold->bias32 && fold->N > 0 && !s2 && !silu), "groupnorm: fold needs a weight, its outputs, one source and no SiLU");
    if (!fold && !pre_part && (C / G) % 2 == 0 && C1 % 2 == 0 && rows * C * 2 <= small_bytes && (long)S * G >= 48) {
        uv_prof_begin(UV_CLS_GROUPNORM, 0.0, 4.0 * (double)rows * C, stream);      // one read from memory (the block's second read comes out of L2) + one write
        if (!sharded_stats) {
            hipLaunchKernelGGL((gn_small_kernel<0>), dim3(G, S), dim3(256), 0, stream, s1, s2, C1, C2, rows_per_stat, G, eps, gamma, beta, silu, out, (float*)nullptr);
            uv_prof_end(stream);
            UV_LAUNCH_CHECK();
            return UV_OK;
        }
        hipLaunchKernelGGL((gn_small_kernel<1>), dim3(G, S), dim3(256), 0, stream, s1, s2, C1, C2, rows_per_stat, G, eps, gamma, beta, silu, out, comm->red);
        UV_LAUNCH_CHECK();
        int rc = comm->allreduce(comm->user, comm->byte_off, S * G * 2);
        if (rc) {
            uv_set_error("groupnorm: all-reduce callback failed (%d)", rc);
            return UV_ERR_STATE;
        }
        int blk;
        const int TRs = gn_geometry(C, &blk);
        int rpa_ = (int)(((long)rows_per_stat * S) / ((long)TRs * 1024));
        rpa_ = rpa_ < 2 ? 2 : (rpa_ > 16 ? 16 : rpa_);
        int nb_ = (rows_per_stat + TRs * rpa_ - 1) / (TRs * rpa_);
        if (nb_ < 1) nb_ = 1;
        const int rpb_ = (rows_per_stat + nb_ - 1) / nb_;
        nb_ = (rows_per_stat + rpb_ - 1) / rpb_;
        hipLaunchKernelGGL(gn_apply_kernel, dim3(nb_, S), dim3(blk), 2 * G * sizeof(float), stream, s1, s2, C1, C2, rows_per_stat, rpb_, G, 1, eps,
                           (long)rows_per_stat * comm->world, comm->red, gamma, beta, silu, out);
        uv_prof_end(stream);
        UV_LAUNCH_CHECK();
        return UV_OK;
    }
    int block;
    const int TR = gn_geometry(C, &block);
    // chunks: <= GN_MAX_CHUNKS per stat unit, 32 rows per thread-row on big tensors; on small ones (frame shards, single-branch calls,
    // the deep levels) fewer rows per block so that the grid still has ~3 blocks per CU — a 2-frame shard of the 64x64 level ran the
    // statistics pass on 129 blocks at 0.9 TB/s (17.5 us for 15.7 MB)
    int rptr = (int)(((long)rows_per_stat * S) / ((long)TR * 768));
    rptr = rptr < 4 ? 4 : (rptr > 32 ? 32 : rptr);
    int nchunk = (rows_per_stat + TR * rptr - 1) / (TR * rptr);
    if (nchunk > GN_MAX_CHUNKS) nchunk = GN_MAX_CHUNKS;
    if (nchunk < 1) nchunk = 1;
    const int rpc = (rows_per_stat + nchunk - 1) / nchunk;
    nchunk = (rows_per_stat + rpc - 1) / rpc;
    // what the launches below move: the apply pass reads and writes the tensor; the statistics pass reads it once more unless the producers left them
    uv_prof_begin(UV_CLS_GROUPNORM, 0.0, ((pre_part ? 0.0 : 2.0) + (fold ? 0.0 : 4.0)) * (double)rows * C, stream);
    size_t lds1 = (size_t)3 * TR * C * sizeof(float);
    UV_REQUIRE(lds1 <= 160 * 1024, "groupnorm: LDS %zu too large", lds1);
    const float* chunk_part = part;
    if (!pre_part) {
        hipLaunchKernelGGL(gn_partial_kernel, dim3(nchunk, S), dim3(block), lds1, stream, s1, s2, C1, C2, rows_per_stat, rpc,
                           G, part);
        UV_LAUNCH_CHECK();
    }
    int rpa = (int)(((long)rows_per_stat * S) / ((long)TR * 1024));       // rows per thread-row of the apply pass: 16 on big tensors
    rpa = rpa < 2 ? 2 : (rpa > 16 ? 16 : rpa);
    int nblk = (rows_per_stat + TR * rpa - 1) / (TR * rpa);
    if (nblk > 2048 / (S > 0 ? 1 : 1)) nblk = 2048;
    if (nblk < 1) nblk = 1;
    const int rpb = (rows_per_stat + nblk - 1) / nblk;
    nblk = (rows_per_stat + rpb - 1) / rpb;
    long count_rows = rows_per_stat;
    // reduce the chunk partials ONCE ([S,nchunk,G,2] -> [S,G,2], stored right behind them) instead of letting each of
    // the ~2000 apply blocks walk all chunks serially (that prologue was ~1/2 of the apply kernel's time)
    float* red = part + (size_t)S * (pre_part ? 0 : nchunk) * G * 2;
    const int SG2 = S * G * 2;
    if (comm && comm->world > 1) red = comm->red;
    if (pre_part) {       // the producing convs / linears left (sum, sumsq) per 16-row fragment and 10-channel sub-group: no pass over the tensor(s)
        hipLaunchKernelGGL(gn_reduce_sub_kernel, dim3(S * G), dim3(256), 0, stream, reinterpret_cast<const float2*>(pre_part),
                           reinterpret_cast<const float2*>(pre_part2), C1 / 10, C2 / 10, rows_per_stat / 16, (C / G) / 10, G, S * G, red);
    } else {
        hipLaunchKernelGGL(gn_reduce_chunks_kernel, dim3((SG2 + 3) / 4), dim3(256), 0, stream, chunk_part, red, nchunk, SG2, G * 2);
    }
    UV_LAUNCH_CHECK();
    if (comm && comm->world > 1) {     // frame shard: sum the partials over ranks (SURVEY §8e coupling 1)
        int rc = comm->allreduce(comm->user, comm->byte_off, SG2);
        if (rc) {
            uv_set_error("groupnorm: all-reduce callback failed (%d)", rc);
            return UV_ERR_STATE;
        }
        count_rows = (long)rows_per_stat * comm->world;
    }
    const float* stats = red;
    if (fold) {
        hipLaunchKernelGGL(gn_fold_linear_kernel, dim3((fold->N + 3) / 4, S), dim3(256), 2 * G * sizeof(float), stream, stats, gamma, beta, G, C, count_rows, eps,
                           fold->W, fold->bias, fold->N, fold->W_out, fold->bias32);
        uv_prof_end(stream);
        UV_LAUNCH_CHECK();
        return UV_OK;
    }
    const int nch_apply = 1;
    hipLaunchKernelGGL(gn_apply_kernel, dim3(nblk, S), dim3(block), 2 * G * sizeof(float), stream, s1, s2, C1, C2,
                       rows_per_stat, rpb, G, nch_apply, eps, count_rows, stats, gamma, beta, silu, out);
    uv_prof_end(stream);
    UV_LAUNCH_CHECK();
    return UV_OK;
}

int uv_launch_layernorm(const half_t* x, long ldx, half_t* y, long ldy, const half_t* gamma, const half_t* beta,
                        long rows, int C, float eps, hipStream_t stream) {
    UV_REQUIRE(C % 8 == 0 && C <= 64 * 8 * 4, "layernorm: C=%d unsupported (multiple of 8, <= 2048)", C);
    const int nch = (C / 8 + 63) / 64;
    dim3 block(256);
    uv_prof_begin(UV_CLS_LAYERNORM, 0.0, 4.0 * (double)rows * C, stream);
    if (nch <= 1) hipLaunchKernelGGL((layernorm_kernel<1, 4>), dim3((unsigned)((rows + 15) / 16)), block, 0, stream, x, ldx, y, ldy, gamma, beta, (int)rows, C, eps);
    else if (nch == 2) hipLaunchKernelGGL((layernorm_kernel<2, 2>), dim3((unsigned)((rows + 7) / 8)), block, 0, stream, x, ldx, y, ldy, gamma, beta, (int)rows, C, eps);
    else hipLaunchKernelGGL((layernorm_kernel<4, 1>), dim3((unsigned)((rows + 3) / 4)), block, 0, stream, x, ldx, y, ldy, gamma, beta, (int)rows, C, eps);
    uv_prof_end(stream);
    UV_LAUNCH_CHECK();
    return UV_OK;
}
