// Optical-flow warp + occlusion test + sliding-window accumulation (uint8 frames, fp32 flows).
//   occlusion:  || ((c + fwd) + bwd) - c ||_2 > thr          (fp32, the reference's operation order, no FMA)
//   warp:       cv2.remap(now, c + fwd, INTER_LINEAR, BORDER_CONSTANT) restated exactly: 5-bit sub-pixel
//               positions, 15-bit weights, (sum + 2^14) >> 15
//   blend:      occluded pixels take the key frame; result accumulated in fp32; window mean stored with
//               float -> uint8 truncation (numpy store semantics)
// Replaces src/cal_optica_flow.py:20-46 and the accumulation of stable_diffusion.py:731-747.
#include "common.h"
#include "kernels.h"

namespace {

__global__ __launch_bounds__(256) void warp_accumulate_kernel(const uint8_t* __restrict__ key, const uint8_t* __restrict__ now,
                                                              const float* __restrict__ fwd, const float* __restrict__ bwd,
                                                              float* __restrict__ acc, int H, int W, float thr) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= (long)H * W) return;
    const int y = (int)(p / W), x = (int)(p % W);
    const float cx = (float)x, cy = (float)y;
    const float fx = fwd[p * 2], fy = fwd[p * 2 + 1];
    const float c1x = __fadd_rn(cx, fx), c1y = __fadd_rn(cy, fy);
    const float ex = __fsub_rn(__fadd_rn(c1x, bwd[p * 2]), cx);
    const float ey = __fsub_rn(__fadd_rn(c1y, bwd[p * 2 + 1]), cy);
    const float err = __fsqrt_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)));
    int o[3];
    if (err > thr) {
        o[0] = key[p * 3];
        o[1] = key[p * 3 + 1];
        o[2] = key[p * 3 + 2];
    } else {
        int sx = __float2int_rn(__fmul_rn(c1x, 32.f));
        int sy = __float2int_rn(__fmul_rn(c1y, 32.f));
        int ix = sx >> 5, iy = sy >> 5;
        ix = ix < -32768 ? -32768 : (ix > 32767 ? 32767 : ix);
        iy = iy < -32768 ? -32768 : (iy > 32767 ? 32767 : iy);
        const int ax = sx & 31, ay = sy & 31;
        const int w00 = (32 - ax) * (32 - ay) * 32, w01 = ax * (32 - ay) * 32, w10 = (32 - ax) * ay * 32, w11 = ax * ay * 32;
        const bool x0 = ix >= 0 && ix < W, x1 = ix + 1 >= 0 && ix + 1 < W;
        const bool y0 = iy >= 0 && iy < H, y1 = iy + 1 >= 0 && iy + 1 < H;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            int v00 = (y0 && x0) ? now[((long)iy * W + ix) * 3 + ch] : 0;
            int v01 = (y0 && x1) ? now[((long)iy * W + ix + 1) * 3 + ch] : 0;
            int v10 = (y1 && x0) ? now[((long)(iy + 1) * W + ix) * 3 + ch] : 0;
            int v11 = (y1 && x1) ? now[((long)(iy + 1) * W + ix + 1) * 3 + ch] : 0;
            int r = (v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11 + (1 << 14)) >> 15;
            o[ch] = r < 0 ? 0 : (r > 255 ? 255 : r);
        }
    }
    acc[p * 3] += (float)o[0];
    acc[p * 3 + 1] += (float)o[1];
    acc[p * 3 + 2] += (float)o[2];
}

// One launch per KEY FRAME (round 4; was 2r warp launches + accumulate + store = 58 launches of ~17 us per 16-frame pass): the window
// mean of stable_diffusion.py:731-747 for key frame `key`, in place in the [F,H,W,3] working copy.  The (up to 2r) neighbours are
// warped exactly as warp_accumulate_kernel does, the terms are added in the reference's order b = -r .. r (sums of <= 9 integers
// <= 255 are exact in fp32 whatever the order), the mean is stored with the same float -> uint8 truncation.  A thread reads the key
// frame only at its own pixel and gathers from OTHER frames, so writing est[key] in place is race-free.
// fl.p[2 s], fl.p[2 s + 1]: forward (key -> now) and backward (now -> key) flow [H][W][2] of the s-th in-clip neighbour, increasing bias.
struct WindowFlows {
    const float* p[16];
};
__global__ __launch_bounds__(256) void warp_window_key_kernel(uint8_t* __restrict__ est, WindowFlows fl, int F, int H, int W, int key, int r, float thr) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    const long HW = (long)H * W;
    if (p >= HW) return;
    const int y = (int)(p / W), x = (int)(p % W);
    const float cx = (float)x, cy = (float)y;
    uint8_t* kf = est + (long)key * HW * 3;
    const int k0 = kf[p * 3], k1 = kf[p * 3 + 1], k2 = kf[p * 3 + 2];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    int weight = 0, slot = 0;
    for (int b = -r; b <= r; ++b) {
        const int nowi = key + b;
        if (nowi < 0 || nowi >= F) continue;
        ++weight;
        if (b == 0) {
            a0 += (float)k0; a1 += (float)k1; a2 += (float)k2;
            continue;
        }
        const float* fwd = fl.p[2 * slot];
        const float* bwd = fl.p[2 * slot + 1];
        ++slot;
        const uint8_t* now = est + (long)nowi * HW * 3;
        const float fx = fwd[p * 2], fy = fwd[p * 2 + 1];
        const float c1x = __fadd_rn(cx, fx), c1y = __fadd_rn(cy, fy);
        const float ex = __fsub_rn(__fadd_rn(c1x, bwd[p * 2]), cx);
        const float ey = __fsub_rn(__fadd_rn(c1y, bwd[p * 2 + 1]), cy);
        const float err = __fsqrt_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)));
        int o[3] = {k0, k1, k2};
        if (!(err > thr)) {
            int sx = __float2int_rn(__fmul_rn(c1x, 32.f));
            int sy = __float2int_rn(__fmul_rn(c1y, 32.f));
            int ix = sx >> 5, iy = sy >> 5;
            ix = ix < -32768 ? -32768 : (ix > 32767 ? 32767 : ix);
            iy = iy < -32768 ? -32768 : (iy > 32767 ? 32767 : iy);
            const int ax = sx & 31, ay = sy & 31;
            const int w00 = (32 - ax) * (32 - ay) * 32, w01 = ax * (32 - ay) * 32, w10 = (32 - ax) * ay * 32, w11 = ax * ay * 32;
            const bool x0 = ix >= 0 && ix < W, x1 = ix + 1 >= 0 && ix + 1 < W;
            const bool y0 = iy >= 0 && iy < H, y1 = iy + 1 >= 0 && iy + 1 < H;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                int v00 = (y0 && x0) ? now[((long)iy * W + ix) * 3 + ch] : 0;
                int v01 = (y0 && x1) ? now[((long)iy * W + ix + 1) * 3 + ch] : 0;
                int v10 = (y1 && x0) ? now[((long)(iy + 1) * W + ix) * 3 + ch] : 0;
                int v11 = (y1 && x1) ? now[((long)(iy + 1) * W + ix + 1) * 3 + ch] : 0;
                int rr = (v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11 + (1 << 14)) >> 15;
                o[ch] = rr < 0 ? 0 : (rr > 255 ? 255 : rr);
            }
        }
        a0 += (float)o[0]; a1 += (float)o[1]; a2 += (float)o[2];
    }
    const float wf = (float)weight;
    const float m[3] = {__fdiv_rn(a0, wf), __fdiv_rn(a1, wf), __fdiv_rn(a2, wf)};
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const int t = (int)m[ch];
        kf[p * 3 + ch] = (uint8_t)(t < 0 ? 0 : (t > 255 ? 255 : t));
    }
}

__global__ void accumulate_u8_kernel(const uint8_t* __restrict__ f, float* __restrict__ acc, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) acc[i] += (float)f[i];
}
__global__ void window_store_kernel(const float* __restrict__ acc, float weight, uint8_t* __restrict__ dst, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        float v = __fdiv_rn(acc[i], weight);
        int t = (int)v;                               // C cast: truncation toward zero
        dst[i] = (uint8_t)(t < 0 ? 0 : (t > 255 ? 255 : t));
    }
}

// ---- latent-space variant of the sliding window (README.md:59 of the reference describes smoothing "in the
// latent space", its code only has the pixel variant, so THIS definition is ours and has no reference oracle):
// one launch per key frame k (Gauss-Seidel like the pixel loop: k sees the already smoothed k-2, k-1):
//   est[k] <- mean over b in [-r, r], 0 <= k+b < F of  (b == 0 ? est[k] : occluded ? est[k] : bilinear(est[k+b], p + fwd))
//   fwd = lflow[k][b+r], bwd = lflow[k+b][r-b] (flows in latent-pixel units, added at the same pixel like
//   cal_optica_flow.py:20-29), occluded iff |fwd + bwd|_2 > thr, bilinear taps outside the latent are zero.
// x0 is [C, F, h, w] fp16 (the pipeline's [1,C,F,h,w] pred_original_sample), updated in place.
__global__ __launch_bounds__(256) void latent_window_kernel(half_t* __restrict__ x0, const float* __restrict__ lflow, int C, int F, int h, int w,
                                                            int r, int key, float thr) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int hw = h * w;
    if (p >= hw) return;
    const int y = p / w, x = p - y * w;
    const int nb = 2 * r + 1;
    float acc[8], kv[8];
    for (int c = 0; c < C; ++c) {
        kv[c] = (float)x0[((long)c * F + key) * hw + p];
        acc[c] = kv[c];
    }
    float weight = 1.f;
    for (int b = -r; b <= r; ++b) {
        const int now = key + b;
        if (b == 0 || now < 0 || now >= F) continue;
        const float* fw = lflow + (((long)key * nb + (b + r)) * hw + p) * 2;
        const float* bw = lflow + (((long)now * nb + (r - b)) * hw + p) * 2;
        const float fx = fw[0], fy = fw[1];
        const float ex = fx + bw[0], ey = fy + bw[1];
        weight += 1.f;
        if (sqrtf(ex * ex + ey * ey) > thr) {
            for (int c = 0; c < C; ++c) acc[c] += kv[c];
            continue;
        }
        const float sx = (float)x + fx, sy = (float)y + fy;
        const float flx = floorf(sx), fly = floorf(sy);
        const int ix = (int)flx, iy = (int)fly;
        const float ax = sx - flx, ay = sy - fly;
        const bool x0ok = ix >= 0 && ix < w, x1ok = ix + 1 >= 0 && ix + 1 < w;
        const bool y0ok = iy >= 0 && iy < h, y1ok = iy + 1 >= 0 && iy + 1 < h;
        const float w00 = (1.f - ax) * (1.f - ay), w01 = ax * (1.f - ay), w10 = (1.f - ax) * ay, w11 = ax * ay;
        for (int c = 0; c < C; ++c) {
            const half_t* src = x0 + ((long)c * F + now) * hw;
            const float v00 = (y0ok && x0ok) ? (float)src[iy * w + ix] : 0.f;
            const float v01 = (y0ok && x1ok) ? (float)src[iy * w + ix + 1] : 0.f;
            const float v10 = (y1ok && x0ok) ? (float)src[(iy + 1) * w + ix] : 0.f;
            const float v11 = (y1ok && x1ok) ? (float)src[(iy + 1) * w + ix + 1] : 0.f;
            acc[c] += v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11;
        }
    }
    for (int c = 0; c < C; ++c) x0[((long)c * F + key) * hw + p] = (half_t)(acc[c] / weight);
}

}  // namespace

int uv_launch_latent_window_smooth(half_t* x0, const float* lflow, int C, int F, int h, int w, int r, float thr, hipStream_t s) {
    UV_REQUIRE(C >= 1 && C <= 8 && F >= 1 && r >= 1 && r <= 4, "latent_window_smooth: C=%d F=%d r=%d unsupported", C, F, r);
    for (int key = 0; key < F; ++key) {       // sequential over key frames: frame k reads the already smoothed k-r .. k-1
        hipLaunchKernelGGL(latent_window_kernel, dim3((unsigned)((h * w + 255) / 256)), dim3(256), 0, s, x0, lflow, C, F, h, w, r, key, thr);
        UV_LAUNCH_CHECK();
    }
    return UV_OK;
}

int uv_launch_warp_accumulate(const uint8_t* key, const uint8_t* now, const float* fwd, const float* bwd, float* acc, int H, int W,
                              float thr, hipStream_t s) {
    hipLaunchKernelGGL(warp_accumulate_kernel, dim3((unsigned)(((long)H * W + 255) / 256)), dim3(256), 0, s, key, now, fwd, bwd, acc,
                       H, W, thr);
    UV_LAUNCH_CHECK();
    return UV_OK;
}
int uv_launch_warp_window_key(uint8_t* est, const float* const* flows, int nn, int F, int H, int W, int key, int r, float thr, hipStream_t s) {
    UV_REQUIRE(est && F >= 1 && key >= 0 && key < F && r >= 1 && r <= 4 && H >= 1 && W >= 1, "warp_window_key: bad geometry (F=%d key=%d r=%d)", F, key, r);
    int want = 0;
    for (int b = -r; b <= r; ++b) want += (b != 0 && key + b >= 0 && key + b < F);
    UV_REQUIRE(nn == want && (nn == 0 || flows), "warp_window_key: key frame %d of %d has %d in-clip neighbours within r = %d, %d flow pairs given", key, F, want, r, nn);
    WindowFlows fl{};
    for (int i = 0; i < 2 * nn; ++i) {
        UV_REQUIRE(flows[i], "warp_window_key: flow pointer %d is null", i);
        fl.p[i] = flows[i];
    }
    hipLaunchKernelGGL(warp_window_key_kernel, dim3((unsigned)(((long)H * W + 255) / 256)), dim3(256), 0, s, est, fl, F, H, W, key, r, thr);
    UV_LAUNCH_CHECK();
    return UV_OK;
}
int uv_launch_accumulate_u8(const uint8_t* f, float* acc, long n, hipStream_t s) {
    hipLaunchKernelGGL(accumulate_u8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, f, acc, n);
    UV_LAUNCH_CHECK();
    return UV_OK;
}
int uv_launch_window_store(const float* acc, float weight, uint8_t* dst, long n, hipStream_t s) {
    hipLaunchKernelGGL(window_store_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, acc, weight, dst, n);
    UV_LAUNCH_CHECK();
    return UV_OK;
}
