// Point-matching mask propagation on the GPU (fp32 like the reference's .float() features).
//   aff[j][i]  = exp( <tar_i/|tar_i|, src_j/|src_j|> / T )            [Nsrc, hw]
//   keep entries >= the top-k-th value of their column, column-normalise, segs_tar = segs_src @ aff
//   finalize: bilinear up to HxW (torch align_corners=False), per-class min-max, FIRST-max argmax, !=0 -> 255
// Replaces src/mask_propagation.py:72-83 (mask_propogation core) and :60-69 (upsample / norm_mask / argmax).
// The random sub-sampling (:87-97, torch.randperm on the host RNG) stays on the host on purpose so that the
// index stream is bit-identical to the reference's.
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

// one thread per target column: k-th largest value (k <= 16) by a register-resident sorted list, then
// zero everything below it, sum, and normalise the column in place.
template <int KMAX>
__global__ __launch_bounds__(64) void topk_normalize_kernel(float* __restrict__ aff, int Nsrc, int hw, int k) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= hw) return;
    float t[KMAX];
#pragma unroll
    for (int q = 0; q < KMAX; ++q) t[q] = -INFINITY;
    for (int j = 0; j < Nsrc; ++j) {
        float v = aff[(long)j * hw + i];
        if (v > t[KMAX - 1]) {
#pragma unroll
            for (int q = 0; q < KMAX; ++q) {
                float hi = fmaxf(t[q], v), lo = fminf(t[q], v);
                t[q] = hi;
                v = lo;
            }
        }
    }
    float thr = t[0];
#pragma unroll
    for (int q = 0; q < KMAX; ++q)
        if (q < k) thr = t[q];                      // k-th largest (sorted descending)
    float s = 0.f;
    for (int j = 0; j < Nsrc; ++j) {
        float v = aff[(long)j * hw + i];
        v = v < thr ? 0.f : v;
        s += v;
    }
    for (int j = 0; j < Nsrc; ++j) {
        float v = aff[(long)j * hw + i];
        v = v < thr ? 0.f : v;
        aff[(long)j * hw + i] = v / s;
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
    hipLaunchKernelGGL((sgemm_kernel<true, true>), dim3((hw + 63) / 64, (Nsrc + 63) / 64), dim3(256), 0, s, sn, tn, aff, Nsrc, hw, C, T);
    hipLaunchKernelGGL((topk_normalize_kernel<16>), dim3((hw + 63) / 64), dim3(64), 0, s, aff, Nsrc, hw, topk);
    hipLaunchKernelGGL((sgemm_kernel<false, false>), dim3((hw + 63) / 64, (ncls + 63) / 64), dim3(256), 0, s, segs_src, aff, segs_tar,
                       ncls, hw, Nsrc, 1.f);
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
