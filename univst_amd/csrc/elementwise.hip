// Small HBM-bound helpers around the UNet: layout conversion (the ONLY place the reference's
// [b,c,f,h,w] layout is touched), timestep sinusoid, DDIM update, mask blend, mask bilinear resize.
// Replaces: the ~450 rearrange/contiguous permutes per step of the reference (resnet.py:57-80,
// attention.py:112-153), diffusers Timesteps (unet_3d_condition.py:359-364), DDIMScheduler.step
// (stable_diffusion.py:761) / next_step (ddim_inversion.py:190-204), mask blend (:687-702).
#include "common.h"
#include "kernels.h"

namespace {

// x [B,Cl,F,HW] fp16 -> y [B*F, HW, CP] fp16 (channels >= Cl zero)
__global__ void ncfhw_to_nhwc_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, int B, int Cl, int F, int HW,
                                     int CP) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)B * F * HW;
    if (i >= total) return;
    int hw = (int)(i % HW);
    long bf = i / HW;
    int f = (int)(bf % F), b = (int)(bf / F);
    for (int c = 0; c < CP; ++c)
        y[i * CP + c] = c < Cl ? x[(((long)b * Cl + c) * F + f) * HW + hw] : (half_t)0.f;
}

// x [B*F, HW, ldx>=Cl] fp16 -> y [B,Cl,F,HW]
__global__ void nhwc_to_ncfhw_kernel(const half_t* __restrict__ x, int ldx, half_t* __restrict__ y, int B, int Cl, int F,
                                     int HW) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)B * F * HW;
    if (i >= total) return;
    int hw = (int)(i % HW);
    long bf = i / HW;
    int f = (int)(bf % F), b = (int)(bf / F);
    for (int c = 0; c < Cl; ++c) y[(((long)b * Cl + c) * F + f) * HW + hw] = x[i * ldx + c];
}

// diffusers get_timestep_embedding(flip_sin_to_cos=True, freq_shift=0): [cos | sin], fp32 math -> fp16
__global__ void timestep_embed_kernel(float t, half_t* __restrict__ out, int B, int dim, int flip, float shift) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    int half = dim / 2;
    if (j >= half) return;
    float ex = (-9.210340371976184f * (float)j) / ((float)half - shift);
    float arg = t * expf(ex);
    float sn = sinf(arg), cs = cosf(arg);
    for (int b = 0; b < B; ++b) {
        out[(long)b * dim + j] = (half_t)(flip ? cs : sn);
        out[(long)b * dim + half + j] = (half_t)(flip ? sn : cs);
    }
}

// out = c_x * x + c_e * eps   (DDIM step / inversion next_step folded to two coefficients)
__global__ void axpby_kernel(const half_t* __restrict__ x, const half_t* __restrict__ e, half_t* __restrict__ out, float cx,
                             float ce, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (half_t)(cx * (float)x[i] + ce * (float)e[i]);
}

// lat[c,f,hw] = (1-m[f,hw]) * a[c,f,hw] + m[f,hw] * b[c,f,hw]
__global__ void mask_blend_kernel(const half_t* __restrict__ a, const half_t* __restrict__ b, const half_t* __restrict__ m,
                                  half_t* __restrict__ out, int Cl, long FHW) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)Cl * FHW) return;
    float mv = m ? (float)m[i % FHW] : 0.f;
    out[i] = (half_t)((1.f - mv) * (float)a[i] + mv * (float)b[i]);
}

// torch F.interpolate(mode='bilinear', align_corners=False) of a {0,1} uint8 mask [F,H,W] -> fp16 [F,h,w]
__global__ void mask_resize_kernel(const uint8_t* __restrict__ mask, half_t* __restrict__ out, int F, int H, int W, int h,
                                   int w) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)F * h * w) return;
    int x = (int)(i % w), y = (int)((i / w) % h), f = (int)(i / ((long)w * h));
    float sy = ((float)y + 0.5f) * ((float)H / (float)h) - 0.5f;
    float sx = ((float)x + 0.5f) * ((float)W / (float)w) - 0.5f;
    sy = sy < 0.f ? 0.f : sy;
    sx = sx < 0.f ? 0.f : sx;
    int y0 = (int)sy, x0 = (int)sx;
    int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    float ly = sy - y0, lx = sx - x0;
    const uint8_t* mp = mask + (long)f * H * W;
    float v = (1.f - ly) * ((1.f - lx) * mp[(long)y0 * W + x0] + lx * mp[(long)y0 * W + x1]) +
              ly * ((1.f - lx) * mp[(long)y1 * W + x0] + lx * mp[(long)y1 * W + x1]);
    out[i] = (half_t)v;
}

__global__ void add_bias_rows_kernel(half_t* __restrict__ x, const half_t* __restrict__ b, long rows, int C) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows * C) x[i] = (half_t)((float)x[i] + (float)b[i % C]);
}

// `width` columns from column `col0` of frame `frame` of every branch -> packed [B, N, width]: the frame-shard halo / broadcast payload.
// Round 6: the payload is the transformer block's HIDDEN rows (col0 = 0, width = C; the receiver projects them to K | V itself — half the
// bytes of the K|V pack, and they exist before the q|k|v GEMM runs)
__global__ void rows_pack_kernel(const half_t* __restrict__ src, long ld, int col0, int width, int N, int B, int fpb, int frame, half_t* __restrict__ dst) {
    const int ch = width / 8;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * N * ch) return;
    int c8 = (int)(i % ch);
    long r = i / ch;
    int n = (int)(r % N), b = (int)(r / N);
    const half_t* from = src + ((long)(b * fpb + frame) * N + n) * ld + col0 + c8 * 8;
    *reinterpret_cast<h8*>(dst + r * width + c8 * 8) = *reinterpret_cast<const h8*>(from);
}

}  // namespace

static inline unsigned nblk(long n, int b) { return (unsigned)((n + b - 1) / b); }

int uv_launch_ncfhw_to_nhwc(const half_t* x, half_t* y, int B, int Cl, int F, int HW, int CP, hipStream_t s) {
    hipLaunchKernelGGL(ncfhw_to_nhwc_kernel, dim3(nblk((long)B * F * HW, 256)), dim3(256), 0, s, x, y, B, Cl, F, HW, CP);
    UV_LAUNCH_CHECK();
    return UV_OK;
}
int uv_launch_nhwc_to_ncfhw(const half_t* x, int ldx, half_t* y, int B, int Cl, int F, int HW, hipStream_t s) {
    hipLaunchKernelGGL(nhwc_to_ncfhw_kernel, dim3(nblk((long)B * F * HW, 256)), dim3(256), 0, s, x, ldx, y, B, Cl, F, HW);
    UV_LAUNCH_CHECK();
    return UV_OK;
}
int uv_launch_timestep_embed(float t, half_t* out, int B, int dim, int flip, float shift, hipStream_t s) {
    hipLaunchKernelGGL(timestep_embed_kernel, dim3(nblk(dim / 2, 64)), dim3(64), 0, s, t, out, B, dim, flip, shift);
    UV_LAUNCH_CHECK();
    return UV_OK;
}
int uv_launch_axpby(const half_t* x, const half_t* e, half_t* out, float cx, float ce, long n, hipStream_t s) {
    hipLaunchKernelGGL(axpby_kernel, dim3(nblk(n, 256)), dim3(256), 0, s, x, e, out, cx, ce, n);
    UV_LAUNCH_CHECK();
    return UV_OK;
}
int uv_launch_mask_blend(const half_t* a, const half_t* b, const half_t* m, half_t* out, int Cl, long FHW, hipStream_t s) {
    hipLaunchKernelGGL(mask_blend_kernel, dim3(nblk((long)Cl * FHW, 256)), dim3(256), 0, s, a, b, m, out, Cl, FHW);
    UV_LAUNCH_CHECK();
    return UV_OK;
}
int uv_launch_mask_resize(const uint8_t* mask, half_t* out, int F, int H, int W, int h, int w, hipStream_t s) {
    hipLaunchKernelGGL(mask_resize_kernel, dim3(nblk((long)F * h * w, 256)), dim3(256), 0, s, mask, out, F, H, W, h, w);
    UV_LAUNCH_CHECK();
    return UV_OK;
}
int uv_launch_add_bias_rows(half_t* x, const half_t* b, long rows, int C, hipStream_t s) {
    hipLaunchKernelGGL(add_bias_rows_kernel, dim3(nblk(rows * C, 256)), dim3(256), 0, s, x, b, rows, C);
    UV_LAUNCH_CHECK();
    return UV_OK;
}

int uv_launch_rows_pack(const half_t* src, long ld, int col0, int width, int N, int B, int fpb, int frame, half_t* dst, hipStream_t s) {
    UV_REQUIRE(width % 8 == 0 && col0 % 8 == 0 && ld % 8 == 0, "rows_pack: columns must come in groups of 8");
    hipLaunchKernelGGL(rows_pack_kernel, dim3(nblk((long)B * N * (width / 8), 256)), dim3(256), 0, s, src, ld, col0, width, N, B, fpb, frame, dst);
    UV_LAUNCH_CHECK();
    return UV_OK;
}
