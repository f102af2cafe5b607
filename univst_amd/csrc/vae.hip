// The temporal VAE behind the pipeline's decode / encode call sites (reference call sites
// backbones/video_diffusion_sd/pipelines/stable_diffusion.py:369-394 decode_latents, :793-818 get_images_from_latents, :820-834 get_latent_image,
// inversion_tools/ddim_inversion.py:28-31,52-55) as ONE host-side graph of the library's gfx950 kernels per call.
//
// The network itself is diffusers' AutoencoderKLTemporalDecoder (the SVD VAE: src/sd/run_*_sd.py:36-42 load it from
// stabilityai/stable-video-diffusion-img2vid) — THIRD-PARTY code that is absent from the reference tree and from both boxes.  Its structure is
// restated here from the published definition (diffusers 0.35: models/autoencoders/autoencoder_kl_temporal_decoder.py, vae.py Encoder,
// unets/unet_3d_blocks.py MidBlockTemporalDecoder / UpBlockTemporalDecoder, resnet.py ResnetBlock2D / TemporalResnetBlock /
// SpatioTemporalResBlock / AlphaBlender, attention_processor.py Attention) with the state-dict keys of that class, so a `vae/` checkpoint loads
// as is; PARITY UNPINNED (the oracle oracle/vae_ref.py is a second restatement of the same definition, not the third-party code).
//
//   encoder  conv_in 3->C0 | 4 x [2 ResnetBlock2D, stride-2 conv (input padded bottom / right only)] | mid: resnet, 1-head attention, resnet |
//            GroupNorm + SiLU + conv_out -> 2*latent | quant_conv 1x1                                      (per frame; no temporal layers)
//   decoder  conv_in latent->C3 | mid: ST-resblock, 1-head attention (head_dim = C3), ST-resblock | 4 x [3 ST-resblocks, nearest x2 + conv] |
//            GroupNorm + SiLU + conv_out -> 3 | time_conv_out: Conv3d (3,1,1) over the frames
//   ST-resblock = ResnetBlock2D (per-frame GroupNorm) -> TemporalResnetBlock (GroupNorm over (C/G, F, H, W), two Conv3d (3,1,1)) -> AlphaBlender
//            (merge "learned", switch_spatial_to_temporal_mix: x = (1 - sigmoid(mix)) * spatial + sigmoid(mix) * temporal)
//
// Kernels: the NHWC implicit-GEMM convs (gemm.hip; the Conv3d (3,1,1) is the 3x1 tap geometry on "image rows = frames, columns = pixels" — no
// wasted taps; nearest x2 folded into the conv addressing; the encoder's asymmetric padding is a tap-geometry parameter), GroupNorm (+SiLU)
// (norm.hip), the linears and — for the single 512-wide head over H/8 x W/8 tokens, which no flash kernel of the library is shaped for — the
// attention as two GEMMs per frame around a row softmax (scores of one frame: N x N fp16, 32 MB at 512 x 512).
#include <math.h>
#include <string.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"
#include "kernels.h"
#include "unet.h"
#include "vae.h"

namespace {

inline unsigned nb(long n) { return (unsigned)((n + 255) / 256); }

__global__ void v_convert_f32_f16_kernel(const float* __restrict__ in, half_t* __restrict__ out, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (half_t)in[i];
}
// [Co][Ci][taps] -> [Co][taps][CiP] (zero padded input channels): 2-D convs (taps = kh*kw) and Conv3d (3,1,1) (taps = 3) alike
__global__ void v_permute_conv_weight_kernel(const half_t* __restrict__ in, half_t* __restrict__ out, int Co, int Ci, int taps, int CiP) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)Co * taps * CiP) return;
    const int c = (int)(i % CiP), t = (int)((i / CiP) % taps), o = (int)(i / ((long)CiP * taps));
    out[i] = c < Ci ? in[((long)o * Ci + c) * taps + t] : (half_t)0.f;
}
// [Co][Ci][3][3] -> tap-inner [Co][Ci/64][9][64] (GemmParams::korder = 1: the nine taps of a 64-channel slab are consecutive k tiles)
__global__ void v_permute_conv_weight_ti_kernel(const half_t* __restrict__ in, half_t* __restrict__ out, int Co, int Ci) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)Co * Ci * 9) return;
    const int j = (int)(i % 64), t = (int)((i / 64) % 9), q = (int)((i / (64 * 9)) % (Ci / 64)), o = (int)(i / ((long)Ci * 9));
    out[i] = in[((long)o * Ci + q * 64 + j) * 9 + t];
}
__global__ void v_scale_kernel(const half_t* __restrict__ in, half_t* __restrict__ out, long n, float f) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (half_t)((float)in[i] * f);
}
// softmax over the rows of S [rows][N] fp16, in place, fp32 arithmetic; one wave per row (N % 8 == 0).  The scores arrive ROUNDED TO fp16: a logit beyond
// 65504 is +inf there and exp(inf - inf) would turn the whole row into NaN (SD-family VAE mid-block activations are large: force_upcast in the stock
// config) — scores are clamped to the fp16 range on load, so such a row degrades to a tie between its saturated keys instead
__global__ __launch_bounds__(256) void v_softmax_rows_kernel(half_t* __restrict__ S, long rows, int N) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    half_t* r = S + row * N;
    float m = -INFINITY;
    for (int c = lane * 8; c < N; c += 512) {
        const h8 v = *reinterpret_cast<const h8*>(r + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) m = fmaxf(m, fminf((float)v[e], 65504.f));
    }
    m = wave_max(m);
    float l = 0.f;
    for (int c = lane * 8; c < N; c += 512) {
        const h8 v = *reinterpret_cast<const h8*>(r + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) l += __expf(fminf((float)v[e], 65504.f) - m);
    }
    l = wave_sum(l);
    const float inv = 1.f / l;
    for (int c = lane * 8; c < N; c += 512) {
        h8 v = *reinterpret_cast<const h8*>(r + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (half_t)(__expf(fminf((float)v[e], 65504.f) - m) * inv);
        *reinterpret_cast<h8*>(r + c) = v;
    }
}
// [R][C] -> [C][R] through a 32 x 33 LDS tile
__global__ __launch_bounds__(256) void v_transpose_kernel(const half_t* __restrict__ in, half_t* __restrict__ out, int R, int C) {
    __shared__ half_t t[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int i = threadIdx.x; i < 1024; i += 256) {
        const int r = i >> 5, c = i & 31;
        t[r][c] = (r0 + r < R && c0 + c < C) ? in[(long)(r0 + r) * C + c0 + c] : (half_t)0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) {
        const int c = i >> 5, r = i & 31;
        if (c0 + c < C && r0 + r < R) out[(long)(c0 + c) * R + r0 + r] = t[r][c];
    }
}
// time_conv_out (Conv3d Cc -> Cc, kernel (3,1,1)) on the NHWC rows of conv_out, written straight into the caller's [B*F, Cc, HW] layout
__global__ __launch_bounds__(256) void v_time_conv_out_kernel(const half_t* __restrict__ x, int ldx, half_t* __restrict__ y, const half_t* __restrict__ w,
                                                              const half_t* __restrict__ bias, int B, int F, long HW, int Cc) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * F * HW) return;
    const long pix = i % HW;
    const long bf = i / HW;
    const int f = (int)(bf % F);
    for (int co = 0; co < Cc; ++co) {
        float acc = (float)bias[co];
        for (int t = 0; t < 3; ++t) {
            const int ff = f + t - 1;
            if (ff < 0 || ff >= F) continue;
            const half_t* xr = x + (i + (long)(t - 1) * HW) * ldx;
            for (int ci = 0; ci < Cc; ++ci) acc += (float)w[((long)co * Cc + ci) * 3 + t] * (float)xr[ci];
        }
        y[(bf * Cc + co) * HW + pix] = (half_t)acc;
    }
}

}  // namespace

Vae::~Vae() {
    for (auto& kv : weights) (void)hipFree(kv.second.ptr);
    for (auto& kv : derived) (void)hipFree(kv.second.ptr);
    if (arena.base) (void)hipFree(arena.base);
}

int Vae::load_tensor(const char* key, const void* dev_ptr, int dtype, const int64_t* shape, int ndim, hipStream_t s) {
    UV_REQUIRE(key && dev_ptr && ndim >= 1 && ndim <= 5, "vae_load_tensor: bad arguments");
    UV_REQUIRE(dtype == UNIVST_F16 || dtype == UNIVST_F32, "vae_load_tensor: dtype %d", dtype);
    long n = 1;
    WTensor t;
    for (int i = 0; i < ndim; ++i) {
        n *= shape[i];
        t.shape.push_back(shape[i]);
    }
    UV_REQUIRE(n > 0, "%s: empty tensor", key);
    UV_HIP(hipMalloc((void**)&t.ptr, (size_t)n * sizeof(half_t)));
    if (dtype == UNIVST_F16) UV_HIP(hipMemcpyAsync(t.ptr, dev_ptr, (size_t)n * sizeof(half_t), hipMemcpyDeviceToDevice, s));
    else hipLaunchKernelGGL(v_convert_f32_f16_kernel, dim3(nb(n)), dim3(256), 0, s, (const float*)dev_ptr, t.ptr, n);
    UV_LAUNCH_CHECK();
    auto it = weights.find(key);
    if (it != weights.end()) {
        UV_HIP(hipStreamSynchronize(s));
        (void)hipFree(it->second.ptr);
    }
    weights[key] = t;
    finalized = false;
    return UV_OK;
}

const WTensor* Vae::find(const std::string& k) const {
    auto it = weights.find(k);
    if (it != weights.end()) return &it->second;
    auto jt = derived.find(k);
    return jt != derived.end() ? &jt->second : nullptr;
}
half_t* Vae::W(const std::string& k) {
    const WTensor* t = find(k);
    if (!t) {
        missing = k;
        return nullptr;
    }
    return t->ptr;
}
int Vae::missing_error() {
    uv_set_error("vae: weight '%s' was never loaded", missing.c_str());
    return UV_ERR_STATE;
}

int Vae::finalize(hipStream_t s) {
    for (auto& kv : derived) (void)hipFree(kv.second.ptr);
    derived.clear();
    mix.clear();
    std::vector<std::string> keys;
    for (auto& kv : weights) keys.push_back(kv.first);
    auto ends = [](const std::string& a, const char* suf) {
        size_t n = strlen(suf);
        return a.size() >= n && a.compare(a.size() - n, n, suf) == 0;
    };
    auto derive = [&](const std::string& k, std::vector<long> shape, half_t** out) {
        long n = 1;
        for (long v : shape) n *= v;
        WTensor t;
        t.shape = shape;
        if (hipMalloc((void**)&t.ptr, (size_t)n * sizeof(half_t)) != hipSuccess) {
            uv_set_error("vae_finalize: out of device memory for %s", k.c_str());
            return UV_ERR_HIP;
        }
        derived[k] = t;
        *out = t.ptr;
        return UV_OK;
    };
    for (const std::string& k : keys) {
        const WTensor& t = weights[k];
        if ((t.shape.size() == 4 || t.shape.size() == 5) && ends(k, ".weight") && k != "decoder.time_conv_out.weight") {
            // Conv2d [Co,Ci,kh,kw] / Conv3d [Co,Ci,3,1,1] -> [Co][taps][CiP]
            const int Co = (int)t.shape[0], Ci = (int)t.shape[1];
            int taps = 1;
            for (size_t d = 2; d < t.shape.size(); ++d) taps *= (int)t.shape[d];
            if (t.shape.size() == 4) UV_REQUIRE((taps == 1 || taps == 9) && t.shape[2] == t.shape[3], "%s: only 1x1 / 3x3 Conv2d", k.c_str());
            else UV_REQUIRE(taps == 3 && t.shape[2] == 3, "%s: only (3,1,1) Conv3d", k.c_str());
            const int CiP = (Ci + 7) / 8 * 8;
            half_t* d;
            int rc = derive(k + "#nhwc", {Co, taps, CiP}, &d);
            if (rc) return rc;
            hipLaunchKernelGGL(v_permute_conv_weight_kernel, dim3(nb((long)Co * taps * CiP)), dim3(256), 0, s, t.ptr, d, Co, Ci, taps, CiP);
            if (t.shape.size() == 4 && taps == 9 && Ci % 64 == 0) {      // tap-inner copy for the 256x320 tile's fast im2col addressing
                half_t* d2;
                rc = derive(k + "#ti", {Co, Ci / 64, 9, 64}, &d2);
                if (rc) return rc;
                hipLaunchKernelGGL(v_permute_conv_weight_ti_kernel, dim3(nb((long)Co * Ci * 9)), dim3(256), 0, s, t.ptr, d2, Co, Ci);
            }
        } else if (ends(k, ".to_q.weight") || ends(k, ".to_q.bias")) {
            // the attention scale 1/sqrt(head_dim) (one head: head_dim = C) rides on the q projection, so the fp16 scores are the scaled ones
            long n = 1;
            for (long v : t.shape) n *= v;
            half_t* d;
            int rc = derive(k + "#qs", t.shape, &d);
            if (rc) return rc;
            const long Cq = t.shape[0];
            hipLaunchKernelGGL(v_scale_kernel, dim3(nb(n)), dim3(256), 0, s, t.ptr, d, n, 1.f / sqrtf((float)Cq));
        }
    }
    UV_LAUNCH_CHECK();
    // AlphaBlender mix factors: one scalar per ST-resblock, read once
    for (const std::string& k : keys) {
        if (!ends(k, ".time_mixer.mix_factor")) continue;
        half_t h;
        UV_HIP(hipMemcpyAsync(&h, weights[k].ptr, sizeof(half_t), hipMemcpyDeviceToHost, s));
        UV_HIP(hipStreamSynchronize(s));
        mix[k.substr(0, k.size() - strlen(".time_mixer.mix_factor"))] = (float)h;
    }
    // The AlphaBlender folded into the temporal block's second conv: blended = alpha * s + (1 - alpha) * (s + h) = s + sigmoid(mix) * h with h = conv2(...) + bias,
    // so conv2's weight and bias are scaled by sigmoid(mix) once and the residual add of its epilogue IS the blend (one pass over three tensors less per block)
    for (auto& kv : mix) {
        const float sig = 1.f / (1.f + expf(-kv.second));
        for (const char* suf : {".temporal_res_block.conv2.weight#nhwc", ".temporal_res_block.conv2.bias"}) {
            const WTensor* t = find(kv.first + suf);
            UV_REQUIRE(t, "%s%s missing", kv.first.c_str(), suf);
            long n = 1;
            for (long v : t->shape) n *= v;
            half_t* d;
            int rc = derive(kv.first + suf + "#mix", t->shape, &d);
            if (rc) return rc;
            hipLaunchKernelGGL(v_scale_kernel, dim3(nb(n)), dim3(256), 0, s, t->ptr, d, n, sig);
        }
    }
    UV_LAUNCH_CHECK();
    UV_HIP(hipStreamSynchronize(s));
    finalized = true;
    return UV_OK;
}

#define RUN(x)                \
    do {                      \
        int _rc = (x);        \
        if (_rc) return _rc;  \
    } while (0)

namespace {
struct VFwd {
    Vae& u;
    hipStream_t s;
    float* gn_ws = nullptr;

    half_t* alloc(long elems) {
        half_t* p = (half_t*)u.arena.alloc((size_t)elems * sizeof(half_t));
        if (!p) uv_set_error("vae: activation arena exhausted (%zu bytes)", u.arena.size);
        return p;
    }
    void free(void* p) { u.arena.release(p); }
    half_t* W(const std::string& k) { return u.W(k); }

    int groupnorm(const Act& a, long rows_per_stat, const std::string& p, int silu, half_t* out, float eps = 1e-6f) {
        half_t *g = W(p + ".weight"), *b = W(p + ".bias");
        if (!g || !b) return u.missing_error();
        return uv_launch_groupnorm(a.p, nullptr, a.C, 0, a.rows(), (int)rows_per_stat, u.cfg.norm_num_groups, eps, g, b, silu, out, gn_ws, s);
    }
    // 2-D conv on [imgs, H, W, C]; asym: the encoder's stride-2 conv (input padded at the bottom / right only)
    int conv(const Act& a, const std::string& p, int Cout, int taps, int stride, int up, bool asym, const half_t* R, Act* out) {
        GemmParams g;
        g.X = a.p;
        g.C1 = a.C;
        g.Hs = a.H;
        g.Ws = a.W;
        g.up = up;
        g.stride = stride;
        g.taps = taps;
        const int He = a.H << up, We = a.W << up;
        if (asym) {
            g.tapw = 3;
            g.pady = g.padx = 0;
            g.Ho = (He + 1 - 3) / stride + 1;
            g.Wo = (We + 1 - 3) / stride + 1;
        } else {
            g.Ho = taps == 9 ? (He + 2 - 3) / stride + 1 : He;
            g.Wo = taps == 9 ? (We + 2 - 3) / stride + 1 : We;
        }
        g.M = a.imgs * g.Ho * g.Wo;
        g.N = Cout;
        g.K = taps * a.C;
        g.korder = (taps == 9 && !asym && stride == 1 && a.C % 64 == 0 && u.find(p + ".weight#ti")) ? 1 : 0;
        g.W = W(p + (g.korder ? ".weight#ti" : ".weight#nhwc"));
        g.bias = W(p + ".bias");
        if (!g.W || !g.bias) return u.missing_error();
        g.R = R;
        g.ldr = Cout;
        out->imgs = a.imgs;
        out->H = g.Ho;
        out->W = g.Wo;
        out->C = Cout;
        out->p = alloc(out->rows() * Cout);
        if (!out->p) return UV_ERR_STATE;
        g.Y = out->p;
        g.ldy = Cout;
        return uv_launch_gemm(g, 1, s);
    }
    // Conv3d (3,1,1) over the F frames of every pixel: the 3x1 tap geometry on (image rows = frames, image columns = pixels)
    int frame_conv(const Act& a, int F, const std::string& p, const half_t* R, half_t* out, const char* mixsuf = "") {
        GemmParams g;
        g.X = a.p;
        g.C1 = a.C;
        g.Hs = F;
        g.Ws = a.H * a.W;
        g.taps = 3;
        g.tapw = 1;
        g.pady = 1;
        g.padx = 0;
        g.Ho = F;
        g.Wo = g.Ws;
        g.M = (int)a.rows();
        g.N = a.C;
        g.K = 3 * a.C;
        g.W = W(p + ".weight#nhwc" + mixsuf);
        g.bias = W(p + ".bias" + mixsuf);
        if (!g.W || !g.bias) return u.missing_error();
        g.R = R;
        g.ldr = a.C;
        g.Y = out;
        g.ldy = a.C;
        return uv_launch_gemm(g, 1, s);
    }
    int linear(const half_t* X, long M, int K, const half_t* Wt, const half_t* bias, int N, half_t* Y, const half_t* R = nullptr) {
        GemmParams g;
        g.X = X;
        g.ldx = K;
        g.M = (int)M;
        g.K = K;
        g.N = N;
        g.W = Wt;
        g.bias = bias;
        g.Y = Y;
        g.ldy = N;
        g.R = R;
        g.ldr = N;
        return uv_launch_gemm(g, 0, s);
    }
    // diffusers ResnetBlock2D without a time embedding: GroupNorm (per image) + SiLU -> conv -> GroupNorm + SiLU -> conv, + (1x1 conv of) the input
    int resnet2d(const std::string& p, const Act& x, int Cout, Act* out) {
        const long hw = (long)x.H * x.W;
        half_t* n1 = alloc(x.rows() * x.C);
        if (!n1) return UV_ERR_STATE;
        RUN(groupnorm(x, hw, p + ".norm1", 1, n1));
        Act n1a{n1, x.imgs, x.H, x.W, x.C}, h;
        RUN(conv(n1a, p + ".conv1", Cout, 9, 1, 0, false, nullptr, &h));
        free(n1);
        half_t* n2 = alloc(h.rows() * Cout);
        if (!n2) return UV_ERR_STATE;
        RUN(groupnorm(h, hw, p + ".norm2", 1, n2));
        free(h.p);
        const half_t* res = x.p;
        Act sc{};
        if (u.find(p + ".conv_shortcut.weight")) {
            RUN(conv(x, p + ".conv_shortcut", Cout, 1, 1, 0, false, nullptr, &sc));
            res = sc.p;
        } else {
            UV_REQUIRE(x.C == Cout, "%s: no conv_shortcut but %d -> %d channels", p.c_str(), x.C, Cout);
        }
        Act n2a{n2, x.imgs, x.H, x.W, Cout};
        RUN(conv(n2a, p + ".conv2", Cout, 9, 1, 0, false, res, out));
        free(n2);
        if (sc.p) free(sc.p);
        return UV_OK;
    }
    // diffusers TemporalResnetBlock (in == out channels, no time embedding) on [B, F, H, W, C]: GroupNorm over (C/G, F, H, W), eps 1e-5
    // (Mid / UpBlockTemporalDecoder build their ST-resblocks with eps = 1e-6, temporal_eps = 1e-5), with the ST-resblock's AlphaBlender in its last epilogue
    int resnet_temporal(const std::string& p, const Act& x, int F, half_t* out) {
        const long rps = (long)F * x.H * x.W;
        half_t* n1 = alloc(x.rows() * x.C);
        if (!n1) return UV_ERR_STATE;
        RUN(groupnorm(x, rps, p + ".norm1", 1, n1, 1e-5f));
        half_t* h = alloc(x.rows() * x.C);
        if (!h) return UV_ERR_STATE;
        Act n1a{n1, x.imgs, x.H, x.W, x.C};
        RUN(frame_conv(n1a, F, p + ".conv1", nullptr, h));
        Act ha{h, x.imgs, x.H, x.W, x.C};
        RUN(groupnorm(ha, rps, p + ".norm2", 1, n1, 1e-5f));
        RUN(frame_conv(n1a, F, p + ".conv2", x.p, out, "#mix"));     // x + sigmoid(mix) * (conv2 + bias): the block's output AND the AlphaBlender (finalize)
        free(h);
        free(n1);
        return UV_OK;
    }
    // diffusers SpatioTemporalResBlock with AlphaBlender("learned", switch_spatial_to_temporal_mix = True)
    int st_resblock(const std::string& p, const Act& x, int F, int Cout, Act* out) {
        Act sp;
        RUN(resnet2d(p + ".spatial_res_block", x, Cout, &sp));
        auto it = u.mix.find(p);
        UV_REQUIRE(it != u.mix.end(), "%s: time_mixer.mix_factor missing", p.c_str());
        half_t* tp = alloc(sp.rows() * Cout);
        if (!tp) return UV_ERR_STATE;
        RUN(resnet_temporal(p + ".temporal_res_block", sp, F, tp));      // = alpha * spatial + (1 - alpha) * temporal, alpha = 1 - sigmoid(mix): folded into conv2
        free(sp.p);
        sp.p = tp;
        *out = sp;
        return UV_OK;
    }
    // diffusers Attention(heads = 1, dim_head = C, norm_num_groups, residual_connection, bias) over the H*W tokens of every image.
    // Numerics: the scaled scores of a frame are stored in fp16 between the QK^T GEMM and the row softmax (|ds| <= 2^-11 |s|: for logits of a few tens
    // that is the size of the error the fp16 q and k rows themselves carry — the reference runs this VAE in fp16 too, `vae.to(weight_dtype)`); the softmax
    // arithmetic and the PV accumulation are fp32.  A d = 512 flash kernel would avoid the N x N round trip; at 64 x 64 tokens the two GEMMs take 2 % of a decode.
    int attention(const std::string& p, const Act& x, Act* out) {
        const int C = x.C, N = x.H * x.W;
        const long rows = x.rows();
        UV_REQUIRE(N % 8 == 0 && C % 8 == 0, "%s: %d tokens x %d channels", p.c_str(), N, C);
        half_t* gn = alloc(rows * C);
        if (!gn) return UV_ERR_STATE;
        RUN(groupnorm(x, N, p + ".group_norm", 0, gn));
        half_t *q = alloc(rows * C), *k = alloc(rows * C), *v = alloc(rows * C);
        if (!q || !k || !v) return UV_ERR_STATE;
        half_t *wq = W(p + ".to_q.weight#qs"), *bq = W(p + ".to_q.bias#qs"), *wk = W(p + ".to_k.weight"), *bk = W(p + ".to_k.bias"), *wv = W(p + ".to_v.weight"),
               *bv = W(p + ".to_v.bias"), *wo = W(p + ".to_out.0.weight"), *bo = W(p + ".to_out.0.bias");
        if (!wq || !bq || !wk || !bk || !wv || !bv || !wo || !bo) return u.missing_error();
        RUN(linear(gn, rows, C, wq, bq, C, q));
        RUN(linear(gn, rows, C, wk, bk, C, k));
        RUN(linear(gn, rows, C, wv, bv, C, v));
        half_t *S = alloc((long)N * N), *vT = alloc((long)N * C);
        if (!S || !vT) return UV_ERR_STATE;
        for (int f = 0; f < x.imgs; ++f) {
            const long o = (long)f * N * C;
            RUN(linear(q + o, N, C, k + o, nullptr, N, S));                              // scores [N, N] (already scaled: the scale rides on q)
            hipLaunchKernelGGL(v_softmax_rows_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, S, (long)N, N);
            hipLaunchKernelGGL(v_transpose_kernel, dim3((C + 31) / 32, (N + 31) / 32), dim3(256), 0, s, v + o, vT, N, C);
            UV_LAUNCH_CHECK();
            RUN(linear(S, N, N, vT, nullptr, C, gn + o));                                 // O = P V  (gn is free by now)
        }
        free(S);
        free(vT);
        free(q);
        free(k);
        free(v);
        out->imgs = x.imgs; out->H = x.H; out->W = x.W; out->C = C;
        out->p = alloc(rows * C);
        if (!out->p) return UV_ERR_STATE;
        RUN(linear(gn, rows, C, wo, bo, C, out->p, x.p));
        free(gn);
        return UV_OK;
    }
};
}  // namespace

int Vae::reserve(long imgs, int H, int Wd) {
    // the largest live set is at the output resolution: ~5 tensors of the second-narrowest width (the last upsampler's output and the GroupNorm copies
    // around it) plus the one-frame score matrix of the attention at 1/8 resolution
    const long rows = imgs * H * Wd;
    const long N = (long)(H / 8) * (Wd / 8);
    size_t need = (size_t)rows * cfg.block_out_channels[1] * 2 * 5 + (size_t)N * N * 2 + (size_t)imgs * N * cfg.block_out_channels[3] * 2 * 8 + (256u << 20);
    if (arena.size < need) {
        UV_HIP(hipDeviceSynchronize());
        if (arena.base) UV_HIP(hipFree(arena.base));
        arena.base = nullptr;
        arena.size = 0;
        UV_HIP(hipMalloc((void**)&arena.base, need));
        arena.size = need;
    }
    arena.reset();
    return UV_OK;
}

// z [imgs, latent, h, w] fp16 (already divided by the scaling factor) -> out [imgs, out_channels, 8h, 8w] fp16; imgs = clips x num_frames
int Vae::decode(const half_t* z, long imgs, int num_frames, int h, int w, half_t* out, hipStream_t s) {
    UV_REQUIRE(finalized, "vae_decode: call univst_vae_finalize after loading weights");
    UV_REQUIRE(imgs > 0 && num_frames > 0 && imgs % num_frames == 0 && h > 0 && w > 0, "vae_decode: %ld images are not whole clips of %d frames", imgs, num_frames);
    const int* boc = cfg.block_out_channels;
    RUN(reserve(imgs, h * 8, w * 8));
    VFwd f{*this, s};
    f.gn_ws = (float*)arena.alloc((size_t)uv_groupnorm_workspace_floats((int)imgs, cfg.norm_num_groups) * 4);
    if (!f.gn_ws) return UV_ERR_STATE;
    const int Lp = (cfg.latent_channels + 7) / 8 * 8;
    Act x{f.alloc(imgs * h * w * Lp), (int)imgs, h, w, Lp};
    if (!x.p) return UV_ERR_STATE;
    RUN(uv_launch_ncfhw_to_nhwc(z, x.p, (int)imgs, cfg.latent_channels, 1, h * w, Lp, s));
    Act cur;
    RUN(f.conv(x, "decoder.conv_in", boc[3], 9, 1, 0, false, nullptr, &cur));
    f.free(x.p);
    Act nxt;
    // mid block: resnets[0], attention, resnets[1]
    RUN(f.st_resblock("decoder.mid_block.resnets.0", cur, num_frames, boc[3], &nxt));
    f.free(cur.p);
    cur = nxt;
    RUN(f.attention("decoder.mid_block.attentions.0", cur, &nxt));
    f.free(cur.p);
    cur = nxt;
    if (cfg.layers_per_block >= 2) {      // (MidBlockTemporalDecoder.forward zips resnets[1:] with its ONE attention: only resnets[1] runs behind it)
        RUN(f.st_resblock("decoder.mid_block.resnets.1", cur, num_frames, boc[3], &nxt));
        f.free(cur.p);
        cur = nxt;
    }
    for (int b = 0; b < 4; ++b) {
        const int Cout = boc[3 - b];
        for (int l = 0; l < cfg.layers_per_block + 1; ++l) {
            RUN(f.st_resblock("decoder.up_blocks." + std::to_string(b) + ".resnets." + std::to_string(l), cur, num_frames, Cout, &nxt));
            f.free(cur.p);
            cur = nxt;
        }
        if (b < 3) {
            RUN(f.conv(cur, "decoder.up_blocks." + std::to_string(b) + ".upsamplers.0.conv", Cout, 9, 1, 1, false, nullptr, &nxt));
            f.free(cur.p);
            cur = nxt;
        }
    }
    half_t* n = f.alloc(cur.rows() * cur.C);
    if (!n) return UV_ERR_STATE;
    RUN(f.groupnorm(cur, (long)cur.H * cur.W, "decoder.conv_norm_out", 1, n));
    Act na{n, cur.imgs, cur.H, cur.W, cur.C};
    f.free(cur.p);
    RUN(f.conv(na, "decoder.conv_out", cfg.out_channels, 9, 1, 0, false, nullptr, &nxt));
    f.free(n);
    half_t *tw = W("decoder.time_conv_out.weight"), *tb = W("decoder.time_conv_out.bias");
    if (!tw || !tb) return missing_error();
    const long HW = (long)nxt.H * nxt.W;
    hipLaunchKernelGGL(v_time_conv_out_kernel, dim3(nb(imgs * HW)), dim3(256), 0, s, nxt.p, cfg.out_channels, out, tw, tb, (int)(imgs / num_frames), num_frames, HW,
                       cfg.out_channels);
    UV_LAUNCH_CHECK();
    f.free(nxt.p);
    return UV_OK;
}

// x [imgs, in_channels, H, W] fp16 in [-1, 1] -> moments [imgs, 2*latent, H/8, W/8] fp16 (mean | logvar; the caller samples)
int Vae::encode(const half_t* xin, long imgs, int H, int Wd, half_t* moments, hipStream_t s) {
    UV_REQUIRE(finalized, "vae_encode: call univst_vae_finalize after loading weights");
    UV_REQUIRE(imgs > 0 && H % 8 == 0 && Wd % 8 == 0, "vae_encode: %ld images of %d x %d (multiples of 8)", imgs, H, Wd);
    const int* boc = cfg.block_out_channels;
    RUN(reserve(imgs, H, Wd));
    VFwd f{*this, s};
    f.gn_ws = (float*)arena.alloc((size_t)uv_groupnorm_workspace_floats((int)imgs, cfg.norm_num_groups) * 4);
    if (!f.gn_ws) return UV_ERR_STATE;
    const int Ip = (cfg.in_channels + 7) / 8 * 8;
    Act x{f.alloc(imgs * H * Wd * Ip), (int)imgs, H, Wd, Ip};
    if (!x.p) return UV_ERR_STATE;
    RUN(uv_launch_ncfhw_to_nhwc(xin, x.p, (int)imgs, cfg.in_channels, 1, H * Wd, Ip, s));
    Act cur, nxt;
    RUN(f.conv(x, "encoder.conv_in", boc[0], 9, 1, 0, false, nullptr, &cur));
    f.free(x.p);
    for (int b = 0; b < 4; ++b) {
        for (int l = 0; l < cfg.layers_per_block; ++l) {
            RUN(f.resnet2d("encoder.down_blocks." + std::to_string(b) + ".resnets." + std::to_string(l), cur, boc[b], &nxt));
            f.free(cur.p);
            cur = nxt;
        }
        if (b < 3) {
            RUN(f.conv(cur, "encoder.down_blocks." + std::to_string(b) + ".downsamplers.0.conv", boc[b], 9, 2, 0, true, nullptr, &nxt));
            f.free(cur.p);
            cur = nxt;
        }
    }
    RUN(f.resnet2d("encoder.mid_block.resnets.0", cur, boc[3], &nxt));
    f.free(cur.p);
    cur = nxt;
    RUN(f.attention("encoder.mid_block.attentions.0", cur, &nxt));
    f.free(cur.p);
    cur = nxt;
    RUN(f.resnet2d("encoder.mid_block.resnets.1", cur, boc[3], &nxt));
    f.free(cur.p);
    cur = nxt;
    half_t* n = f.alloc(cur.rows() * cur.C);
    if (!n) return UV_ERR_STATE;
    RUN(f.groupnorm(cur, (long)cur.H * cur.W, "encoder.conv_norm_out", 1, n));
    Act na{n, cur.imgs, cur.H, cur.W, cur.C};
    f.free(cur.p);
    RUN(f.conv(na, "encoder.conv_out", 2 * cfg.latent_channels, 9, 1, 0, false, nullptr, &cur));
    f.free(n);
    RUN(f.conv(cur, "quant_conv", 2 * cfg.latent_channels, 1, 1, 0, false, nullptr, &nxt));
    f.free(cur.p);
    RUN(uv_launch_nhwc_to_ncfhw(nxt.p, 2 * cfg.latent_channels, moments, (int)imgs, 2 * cfg.latent_channels, 1, nxt.H * nxt.W, s));
    f.free(nxt.p);
    return UV_OK;
}
