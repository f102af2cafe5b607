// Point-matching mask propagation on the GPU (fp32 like the reference's .float() features).
//   aff[j][i]  = exp( <tar_i/|tar_i|, src_j/|src_j|> / T )            [Nsrc, hw]
//   keep entries >= the top-k-th value of their column, column-normalise, segs_tar = segs_src @ aff
//   finalize: bilinear up to HxW (torch align_corners=False), per-class min-max, FIRST-max argmax, !=0 -> 255
// Replaces src/mask_propagation.py:72-83 (mask_propogation core) and :60-69 (upsample / norm_mask / argmax).
// The random sub-sampling (:87-97, torch.randperm on the host RNG) stays on the host on purpose so that the
// index stream is bit-identical to the reference's.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace {

__global__ void rownorm_kernel(const float* __restrict__ x, float* __restrict__ y, int rows, int C) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) {
        float v = x[r * C + c];
        s += v * v;
    }
    s = wave_sum(s);
    const float d = fmaxf(sqrtf(s), 1e-12f);       // F.normalize: x / max(||x||_2, eps)
    for (int c = lane; c < C; c += 64) y[r * C + c] = x[r * C + c] / d;
}

// C[m][n] = sum_k A[m][k] * (TRANSB ? B[n][k] : B[k][n]); EXP: C = exp(C / T).  64x64 tile, 4x4 per thread.
template <bool TRANSB, bool EXP>
__global__ __launch_bounds__(256) void sgemm_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                    float* __restrict__ Cm, int M, int N, int K, float T) {
    __shared__ float As[16][64 + 4];
    __shared__ float Bs[16][64 + 4];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += 16) {
        for (int e = threadIdx.x; e < 64 * 16; e += 256) {
            int mm = e / 16, kk = e % 16;                       // A row-major [M,K]
            int gm = m0 + mm, gk = k0 + kk;
            As[kk][mm] = (gm < M && gk < K) ? A[(long)gm * K + gk] : 0.f;
            if (TRANSB) {
                int gn = n0 + mm;
                Bs[kk][mm] = (gn < N && gk < K) ? B[(long)gn * K + gk] : 0.f;
            }
        }
        if (!TRANSB) {
            for (int e = threadIdx.x; e < 64 * 16; e += 256) {
                int kk = e / 64, nn = e % 64;
                int gk = k0 + kk, gn = n0 + nn;
                Bs[kk][nn] = (gk < K && gn < N) ? B[(long)gk * N + gn] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[kk][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int gm = m0 + ty * 4 + i, gn = n0 + tx * 4 + j;
            if (gm < M && gn < N) Cm[(long)gm * N + gn] = EXP ? expf(acc[i][j] / T) : acc[i][j];
        }
}

// The same GEMM on the fp32 matrix cores (round 3): v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fmaf chain (guide §3
// "FP32-input MFMA": one rounding per product, no wider accumulation), i.e. EXACTLY what sgemm_kernel's inner loop computes, at
// 64 FLOP/clk/SIMD next to an idle VALU — the masks stay bit-identical to the VALU kernel's by construction (k ascending, zero
// padded tails add +0).  Block = 4 waves (2 x 2), wave tile = WM x WN MFMA tiles of 32 x 32; K in 16-wide LDS tiles, k-major
// ([k][m]: a fragment read is 32 consecutive floats per k -> conflict-free ds_read_b32).
typedef float f16v __attribute__((ext_vector_type(16)));
template <bool TRANSB, bool EXP, int WM, int WN>
__global__ __launch_bounds__(256) void sgemm_mfma_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                         float* __restrict__ Cm, int M, int N, int K, float T) {
    constexpr int BM = 64 * WM, BN = 64 * WN, BK = 32;
    __shared__ float As[BK][BM + 4];
    __shared__ float Bs[BK][BN + 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int l31 = lane & 31, kh = lane >> 5;
    const bool vec = (K & 3) == 0 && ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0;      // 16-byte loads along k
    f16v acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int k0 = 0; k0 < K; k0 += BK) {
        if (vec) {                                               // k-contiguous operands: float4 along k, transposed into [k][row]
            for (int e = tid; e < BM * (BK / 4); e += 256) {
                const int mm = e / (BK / 4), k4 = (e % (BK / 4)) * 4;
                const int gm = m0 + mm, gk = k0 + k4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gm < M && gk < K) v = *reinterpret_cast<const float4*>(A + (long)gm * K + gk);
                As[k4][mm] = v.x; As[k4 + 1][mm] = v.y; As[k4 + 2][mm] = v.z; As[k4 + 3][mm] = v.w;
            }
        } else {
            for (int e = tid; e < BM * BK; e += 256) {
                const int mm = e / BK, kk = e % BK;
                const int gm = m0 + mm, gk = k0 + kk;
                As[kk][mm] = (gm < M && gk < K) ? A[(long)gm * K + gk] : 0.f;
            }
        }
        if (TRANSB) {
            if (vec) {
                for (int e = tid; e < BN * (BK / 4); e += 256) {
                    const int nn = e / (BK / 4), k4 = (e % (BK / 4)) * 4;
                    const int gn = n0 + nn, gk = k0 + k4;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (gn < N && gk < K) v = *reinterpret_cast<const float4*>(B + (long)gn * K + gk);
                    Bs[k4][nn] = v.x; Bs[k4 + 1][nn] = v.y; Bs[k4 + 2][nn] = v.z; Bs[k4 + 3][nn] = v.w;
                }
            } else {
                for (int e = tid; e < BN * BK; e += 256) {
                    const int nn = e / BK, kk = e % BK;
                    const int gn = n0 + nn, gk = k0 + kk;
                    Bs[kk][nn] = (gn < N && gk < K) ? B[(long)gn * K + gk] : 0.f;
                }
            }
        } else {
            for (int e = tid; e < BN * BK; e += 256) {
                const int kk = e / BN, nn = e % BN;
                const int gk = k0 + kk, gn = n0 + nn;
                Bs[kk][nn] = (gk < K && gn < N) ? B[(long)gk * N + gn] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[WM], b[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) a[i] = As[kk + kh][(wm * WM + i) * 32 + l31];
#pragma unroll
            for (int j = 0; j < WN; ++j) b[j] = Bs[kk + kh][(wn * WN + j) * 32 + l31];
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = m0 + (wm * WM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                const int gn = n0 + (wn * WN + j) * 32 + l31;
                if (gm < M && gn < N) Cm[(long)gm * N + gn] = EXP ? expf(acc[i][j][r] / T) : acc[i][j][r];
            }
}

// k-th largest value per target column (k <= 16), zero everything below it, column-normalise.
// A block owns 64 columns; its 16 waves walk the source rows in 16 phases (wave w: rows w, w+16, ...), so a wave instruction reads
// 256 contiguous bytes of one row.  Each thread keeps the 16 largest values of its phase in a sorted register list; the 16 lists of
// a column are merged through LDS (the k-th largest value of a set does not depend on the order it is scanned in: exact), the
// kept values are summed per phase and the 16 partial sums added in phase order (fixed order: run-to-run deterministic).
// Round 2 had ONE thread per column walk all Nsrc rows three times (64 waves on the whole chip): 8.2 ms of the 10.2 ms per frame.
//
// Output: the column's surviving entries as a COMPACT list (cnt[i], idx[i][CAP] ascending, val[i][CAP] = v / sum) — after the
// threshold a column of 4096..13.7k affinities has topk (+ ties) non-zeros, and the label product only needs those (sparse_label_kernel).
// A column with more than CAP survivors (a mass tie at the threshold) is written back densely, normalised in place, and flagged
// cnt = -1: the label kernel then walks that column of `aff` like the dense GEMM did.
constexpr int MP_CAP = 32;
template <int KMAX>
__global__ __launch_bounds__(1024) void topk_normalize_kernel(float* __restrict__ aff, int Nsrc, int hw, int k, int* __restrict__ cnt,
                                                              int* __restrict__ idx, float* __restrict__ val) {
    constexpr int NP = 16;
    __shared__ float lists[NP][KMAX][64];      // 64 KB; reused for the survivor lists: int [64][MP_CAP] | float [64][MP_CAP]
    __shared__ float thr_s[64], sum_s[NP][64];
    __shared__ int cnt_s[64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + tx;
    const bool ok = i < hw;
    float t[KMAX];
#pragma unroll
    for (int q = 0; q < KMAX; ++q) t[q] = -INFINITY;
    if (ok) {
        for (int j = ty; j < Nsrc; j += NP) {
            float v = aff[(long)j * hw + i];
            if (v > t[KMAX - 1]) {
#pragma unroll
                for (int q = 0; q < KMAX; ++q) {
                    const float hi = fmaxf(t[q], v), lo = fminf(t[q], v);
                    t[q] = hi;
                    v = lo;
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < KMAX; ++q) lists[ty][q][tx] = t[q];
    __syncthreads();
    if (ty == 0) {
#pragma unroll 1
        for (int p = 1; p < NP; ++p)
#pragma unroll 1
            for (int q = 0; q < KMAX; ++q) {
                float v = lists[p][q][tx];
                if (v > t[KMAX - 1]) {
#pragma unroll
                    for (int r = 0; r < KMAX; ++r) {
                        const float hi = fmaxf(t[r], v), lo = fminf(t[r], v);
                        t[r] = hi;
                        v = lo;
                    }
                } else {
                    break;                    // the lists are sorted descending: nothing further in this one can enter
                }
            }
        float thr = t[0];
#pragma unroll
        for (int q = 0; q < KMAX; ++q)
            if (q < k) thr = t[q];            // k-th largest (sorted descending)
        thr_s[tx] = thr;
        cnt_s[tx] = 0;
    }
    __syncthreads();
    int* sidx = reinterpret_cast<int*>(&lists[0][0][0]);                 // [64][MP_CAP]
    float* sval = reinterpret_cast<float*>(&lists[0][0][0]) + 64 * MP_CAP;
    const float thr = thr_s[tx];
    float s = 0.f;
    if (ok) {
        for (int j = ty; j < Nsrc; j += NP) {
            float v = aff[(long)j * hw + i];
            if (!(v < thr)) {                 // survivor (the reference zeroes v < thr)
                s += v;
                const int slot = atomicAdd(&cnt_s[tx], 1);
                if (slot < MP_CAP) {
                    sidx[tx * MP_CAP + slot] = j;
                    sval[tx * MP_CAP + slot] = v;
                }
            }
        }
    }
    sum_s[ty][tx] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) tot += sum_s[p][tx];
    const int n = cnt_s[tx];
    if (ok && n > MP_CAP) {                   // mass tie: dense fallback for this column
        for (int j = ty; j < Nsrc; j += NP) {
            float v = aff[(long)j * hw + i];
            v = v < thr ? 0.f : v;
            aff[(long)j * hw + i] = v / tot;
        }
        if (ty == 0) cnt[i] = -1;
    } else if (ok && ty == 0) {               // sort the <= MP_CAP survivors by source row (ascending: the dense GEMM's k order)
        int* li = sidx + tx * MP_CAP;
        float* lv = sval + tx * MP_CAP;
        for (int a_ = 1; a_ < n; ++a_) {
            const int kj = li[a_];
            const float kv = lv[a_];
            int b_ = a_ - 1;
            while (b_ >= 0 && li[b_] > kj) {
                li[b_ + 1] = li[b_];
                lv[b_ + 1] = lv[b_];
                --b_;
            }
            li[b_ + 1] = kj;
            lv[b_ + 1] = kv;
        }
        cnt[i] = n;
        for (int e = 0; e < n; ++e) {
            idx[(long)i * MP_CAP + e] = li[e];
            val[(long)i * MP_CAP + e] = lv[e] / tot;
        }
    }
}

// A/B path only (UNIVST_MASKPROP_MFMA=0): rebuild the dense normalised column from the survivor list for the dense label GEMM
__global__ void densify_kernel(float* __restrict__ aff, int Nsrc, int hw, const int* __restrict__ cnt, const int* __restrict__ idx,
                               const float* __restrict__ val) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= hw || cnt[i] < 0) return;
    const int n = cnt[i];
    for (int j = blockIdx.y; j < Nsrc; j += gridDim.y) {
        float v = 0.f;
        for (int e = 0; e < n; ++e)
            if (idx[(long)i * MP_CAP + e] == j) v = val[(long)i * MP_CAP + e];
        aff[(long)j * hw + i] = v;
    }
}

// [ncls][Nsrc] -> [Nsrc][ncls] (32 x 32 LDS tiles): a source row's class vector becomes 1 KB of contiguous memory
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int Cc) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8)
        if (r0 + r < R && c0 + tx < Cc) tile[r][tx] = in[(long)(r0 + r) * Cc + c0 + tx];
    __syncthreads();
    for (int c = ty; c < 32; c += 8)
        if (c0 + c < Cc && r0 + tx < R) out[(long)(c0 + c) * R + r0 + tx] = tile[tx][c];
}

// segs_tar[c][i] = sum_j segs_src[c][j] * aff[j][i] over the SURVIVORS of column i only, as the same ascending-j fmaf chain the
// dense product runs (fma(a, 0, acc) == acc: dropping the zero terms changes no bit).  Block = 64 columns, a wave does 16 of them
// one after the other with its lanes over the classes (coalesced 256-B reads of segsT rows); the 256 x 64 result tile leaves
// through LDS so that a class row is written as 64 consecutive floats.
__global__ __launch_bounds__(256) void sparse_label_kernel(const float* __restrict__ segsT, const float* __restrict__ segs_src,
                                                           const float* __restrict__ aff, const int* __restrict__ cnt,
                                                           const int* __restrict__ idx, const float* __restrict__ val,
                                                           float* __restrict__ segs_tar, int ncls, int hw, int Nsrc) {
    extern __shared__ float outt[];           // [ncls][65]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = blockIdx.x * 64;
    for (int cc = 0; cc < 16; ++cc) {
        const int col = wave * 16 + cc, i = i0 + col;
        if (i >= hw) break;
        const int n = cnt[i];
        for (int c = lane; c < ncls; c += 64) {
            float acc = 0.f;
            if (n >= 0) {
                for (int e = 0; e < n; ++e) acc = fmaf(segsT[(long)idx[(long)i * MP_CAP + e] * ncls + c], val[(long)i * MP_CAP + e], acc);
            } else {                          // dense fallback column
                for (int j = 0; j < Nsrc; ++j) acc = fmaf(segs_src[(long)c * Nsrc + j], aff[(long)j * hw + i], acc);
            }
            outt[c * 65 + col] = acc;
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < ncls * 64; e += 256) {
        const int c = e >> 6, col = e & 63;
        if (i0 + col < hw) segs_tar[(long)c * hw + i0 + col] = outt[c * 65 + col];
    }
}

__device__ __forceinline__ float bilinear_at(const float* __restrict__ src, int h, int w, int H, int W, int y, int x) {
    float sy = ((float)y + 0.5f) * ((float)h / (float)H) - 0.5f;
    float sx = ((float)x + 0.5f) * ((float)w / (float)W) - 0.5f;
    sy = sy < 0.f ? 0.f : sy;
    sx = sx < 0.f ? 0.f : sx;
    int y0 = (int)sy, x0 = (int)sx;
    int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    float ly = sy - y0, lx = sx - x0, hy = 1.f - ly, hx = 1.f - lx;
    return hy * (hx * src[y0 * w + x0] + lx * src[y0 * w + x1]) + ly * (hx * src[y1 * w + x0] + lx * src[y1 * w + x1]);
}

// partial min/max of the up-sampled class field: grid (nchunk, ncls)
__global__ __launch_bounds__(256) void upsample_minmax_kernel(const float* __restrict__ segs, int h, int w, int H, int W,
                                                              float* __restrict__ part) {
    __shared__ float smin[4], smax[4];
    const int c = blockIdx.y, nchunk = gridDim.x;
    const float* src = segs + (long)c * h * w;
    float mn = INFINITY, mx = -INFINITY;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < (long)H * W; p += (long)nchunk * 256) {
        float v = bilinear_at(src, h, w, H, W, (int)(p / W), (int)(p % W));
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    mx = wave_max(mx);
    mn = -wave_max(-mn);
    if ((threadIdx.x & 63) == 0) {
        smin[threadIdx.x >> 6] = mn;
        smax[threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) {
            mn = fminf(mn, smin[i]);
            mx = fmaxf(mx, smax[i]);
        }
        part[((long)c * nchunk + blockIdx.x) * 2] = fminf(mn, smin[0]);
        part[((long)c * nchunk + blockIdx.x) * 2 + 1] = fmaxf(mx, smax[0]);
    }
}

__global__ void reduce_minmax_kernel(const float* __restrict__ part, int nchunk, float* __restrict__ mm, int ncls) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncls) return;
    float mn = INFINITY, mx = -INFINITY;
    for (int i = 0; i < nchunk; ++i) {
        mn = fminf(mn, part[((long)c * nchunk + i) * 2]);
        mx = fmaxf(mx, part[((long)c * nchunk + i) * 2 + 1]);
    }
    mm[c * 2] = mn;
    mm[c * 2 + 1] = mx;
}

__global__ __launch_bounds__(256) void argmax_mask_kernel(const float* __restrict__ segs, const float* __restrict__ mm, int ncls,
                                                          int h, int w, int H, int W, uint8_t* __restrict__ out) {
    long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= (long)H * W) return;
    const int y = (int)(p / W), x = (int)(p % W);
    float best = -INFINITY;
    int bi = 0;
    for (int c = 0; c < ncls; ++c) {
        float v = bilinear_at(segs + (long)c * h * w, h, w, H, W, y, x);
        const float mn = mm[c * 2], mx = mm[c * 2 + 1];
        if (mx > 0.f) {                       // norm_mask: only classes whose max > 0 are rescaled
            v = v - mn;
            v = v / (mx - mn);
        }
        if (v > best) {                       // strict '>' keeps the FIRST maximum (torch.max on CPU)
            best = v;
            bi = c;
        }
    }
    out[p] = bi != 0 ? 255 : 0;
}

}  // namespace

static inline size_t al(size_t x) { return (x + 255) & ~size_t(255); }

int64_t uv_maskprop_workspace_bytes(int hw, int Nsrc, int C) {
    return (int64_t)(al((size_t)hw * C * 4) + al((size_t)Nsrc * C * 4) + al((size_t)Nsrc * hw * 4) + 262144 + 4096);
}

int uv_launch_maskprop_frame(const float* feat_tar, const float* feat_src, const float* segs_src, float* segs_tar, int hw,
                             int Nsrc, int C, int ncls, float T, int topk, void* ws, hipStream_t s) {
    UV_REQUIRE(topk >= 1 && topk <= 16 && topk <= Nsrc, "maskprop: topk=%d must be in 1..16 and <= Nsrc", topk);
    char* w = (char*)ws;
    float* tn = (float*)w;
    w += al((size_t)hw * C * 4);
    float* sn = (float*)w;
    w += al((size_t)Nsrc * C * 4);
    float* aff = (float*)w;
    hipLaunchKernelGGL(rownorm_kernel, dim3((hw + 3) / 4), dim3(256), 0, s, feat_tar, tn, hw, C);
    hipLaunchKernelGGL(rownorm_kernel, dim3((Nsrc + 3) / 4), dim3(256), 0, s, feat_src, sn, Nsrc, C);
    // UNIVST_MASKPROP_MFMA=0 (A/B aid): the fp32 VALU SGEMMs and the dense label product of rounds 1-2 (bit-identical results)
    static const int mfma = getenv("UNIVST_MASKPROP_MFMA") ? atoi(getenv("UNIVST_MASKPROP_MFMA")) : 1;
    if (mfma) hipLaunchKernelGGL((sgemm_mfma_kernel<true, true, 2, 2>), dim3((hw + 127) / 128, (Nsrc + 127) / 128), dim3(256), 0, s, sn, tn, aff, Nsrc, hw, C, T);
    else hipLaunchKernelGGL((sgemm_kernel<true, true>), dim3((hw + 63) / 64, (Nsrc + 63) / 64), dim3(256), 0, s, sn, tn, aff, Nsrc, hw, C, T);
    // stream-ordered scratch for the survivor lists and the transposed labels (their size depends on ncls, which the public
    // workspace query does not take)
    char* sc = nullptr;
    const size_t b_cnt = al((size_t)hw * 4), b_idx = al((size_t)hw * MP_CAP * 4), b_val = b_idx, b_t = al((size_t)Nsrc * ncls * 4);
    UV_HIP(hipMallocAsync((void**)&sc, b_cnt + b_idx + b_val + b_t, s));
    int* cnt = (int*)sc;
    int* idx = (int*)(sc + b_cnt);
    float* val = (float*)(sc + b_cnt + b_idx);
    float* segsT = (float*)(sc + b_cnt + b_idx + b_val);
    hipLaunchKernelGGL((topk_normalize_kernel<16>), dim3((hw + 63) / 64), dim3(1024), 0, s, aff, Nsrc, hw, topk, cnt, idx, val);
    if (mfma) {
        UV_REQUIRE((size_t)ncls * 65 * 4 <= 160 * 1024, "maskprop: ncls=%d too large for the label tile", ncls);
        hipLaunchKernelGGL(transpose_kernel, dim3((Nsrc + 31) / 32, (ncls + 31) / 32), dim3(256), 0, s, segs_src, segsT, ncls, Nsrc);
        hipLaunchKernelGGL(sparse_label_kernel, dim3((hw + 63) / 64), dim3(256), (size_t)ncls * 65 * 4, s, segsT, segs_src, aff, cnt, idx, val, segs_tar,
                           ncls, hw, Nsrc);
    } else {
        hipLaunchKernelGGL(densify_kernel, dim3((hw + 63) / 64, 64), dim3(64), 0, s, aff, Nsrc, hw, cnt, idx, val);
        hipLaunchKernelGGL((sgemm_kernel<false, false>), dim3((hw + 63) / 64, (ncls + 63) / 64), dim3(256), 0, s, segs_src, aff, segs_tar,
                           ncls, hw, Nsrc, 1.f);
    }
    (void)hipFreeAsync(sc, s);
    UV_LAUNCH_CHECK();
    return UV_OK;
}

int uv_launch_maskprop_finalize(const float* segs, uint8_t* out, int ncls, int h, int w, int H, int W, void* ws, hipStream_t s) {
    const int nchunk = 64;
    float* part = (float*)ws;
    float* mm = part + (size_t)ncls * nchunk * 2;
    hipLaunchKernelGGL(upsample_minmax_kernel, dim3(nchunk, ncls), dim3(256), 0, s, segs, h, w, H, W, part);
    hipLaunchKernelGGL(reduce_minmax_kernel, dim3((ncls + 63) / 64), dim3(64), 0, s, part, nchunk, mm, ncls);
    hipLaunchKernelGGL(argmax_mask_kernel, dim3((unsigned)(((long)H * W + 255) / 256)), dim3(256), 0, s, segs, mm, ncls, h, w, H, W, out);
    UV_LAUNCH_CHECK();
    return UV_OK;
}
