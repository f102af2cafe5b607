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

}  // namespace

int uv_launch_warp_accumulate(const uint8_t* key, const uint8_t* now, const float* fwd, const float* bwd, float* acc, int H, int W,
                              float thr, hipStream_t s) {
    hipLaunchKernelGGL(warp_accumulate_kernel, dim3((unsigned)(((long)H * W + 255) / 256)), dim3(256), 0, s, key, now, fwd, bwd, acc,
                       H, W, thr);
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
