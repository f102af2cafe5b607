// The whole pseudo-3D SD UNet forward as ONE host-side graph of gfx950 kernels (one C-ABI call per
// denoising step).  Activations live in NHWC fp16 ([B*F*H*W, C] == the transformer's token layout), chosen
// once at conv_in, so every rearrange/permute/cat/upsample of the reference disappears into kernel
// addressing.  Weights are owned by the handle (device fp16 copies keyed by the reference state-dict
// names) plus derived layouts built by finalize().  A first-fit arena supplies all activations; nothing is
// allocated inside forward() after the first call for a geometry.
//
// Structure restated from: unet_3d_condition.py:306-443, unet_3d_blocks.py:129-645, attention.py:104-346,
// resnet.py:335-394, pnp_utils.py:20-111.  Dead work of the reference that is skipped because it is an
// exact identity for 2-D-initialised weights (verified at finalize): temporal conv1d (dirac, resnet.py:53-55)
// and temporal attention (zero to_out weight => adds its bias, attention.py:233).
#include <math.h>
#include <string.h>

#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"
#include "kernels.h"
#include "unet.h"

namespace {

__global__ void convert_f32_f16_kernel(const float* __restrict__ in, half_t* __restrict__ out, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (half_t)in[i];
}
// [Co,Ci,kh,kw] -> [Co,kh*kw,CiP] (zero padded channels)
__global__ void permute_conv_weight_kernel(const half_t* __restrict__ in, half_t* __restrict__ out, int Co, int Ci, int taps,
                                           int CiP) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)Co * taps * CiP;
    if (i >= total) return;
    int c = (int)(i % CiP);
    int t = (int)((i / CiP) % taps);
    int o = (int)(i / ((long)CiP * taps));
    out[i] = c < Ci ? in[((long)o * Ci + c) * taps + t] : (half_t)0.f;
}
// [Co,Ci,3,3] -> tap-inner [Co][Ci/64][9][64]
__global__ void permute_conv_weight_ti_kernel(const half_t* __restrict__ in, half_t* __restrict__ out, int Co, int Ci) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)Co * Ci * 9;
    if (i >= total) return;
    int j = (int)(i % 64);
    int t = (int)((i / 64) % 9);
    int q = (int)((i / (64 * 9)) % (Ci / 64));
    int o = (int)(i / ((long)Ci * 9));
    out[i] = in[((long)o * Ci + q * 64 + j) * 9 + t];
}
// [Co,Ci,3,3] -> [Co][Ci/32][9][32] (conv_patch_kernel: a k tile is two consecutive (32-channel slab, tap) units)
__global__ void permute_conv_weight_t32_kernel(const half_t* __restrict__ in, half_t* __restrict__ out, int Co, int Ci) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)Co * Ci * 9;
    if (i >= total) return;
    int j = (int)(i % 32);
    int t = (int)((i / 32) % 9);
    int q = (int)((i / (32 * 9)) % (Ci / 32));
    int o = (int)(i / ((long)Ci * 9));
    out[i] = in[((long)o * Ci + q * 32 + j) * 9 + t];
}
// GEGLU row interleave: out row (32q + j) = in row (16q + j), out row (32q+16+j) = in row (half + 16q + j)
__global__ void geglu_interleave_kernel(const half_t* __restrict__ in, half_t* __restrict__ out, int rows, int cols) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)rows * cols) return;
    int c = (int)(i % cols), r = (int)(i / cols);
    int q = r / 32, j = r % 32;
    int src = j < 16 ? 16 * q + j : rows / 2 + 16 * q + (j - 16);
    out[i] = in[(long)src * cols + c];
}
// out[i] = fp16(in[i] * f): to_q weights with log2(e)/sqrt(head_dim) folded in (attention.hip, FOLD kernel)
__global__ void scale_f16_kernel(const half_t* __restrict__ in, half_t* __restrict__ out, long n, float f) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (half_t)((float)in[i] * f);
}

// LayerNorm folded into the following linear (GemmParams::ln_stats): W'[n][k] = fp16(gamma[k] * W[n][k]), wsum[n] = sum_k W'[n][k]
// (of the ROUNDED values: it has to cancel mean * sum_k W' exactly), lnb[n] = bias[n] + sum_k beta[k] * W[n][k].  One wave per row.
__global__ __launch_bounds__(64) void ln_fold_weight_kernel(const half_t* __restrict__ W, const half_t* __restrict__ gamma,
                                                            const half_t* __restrict__ beta, const half_t* __restrict__ bias,
                                                            half_t* __restrict__ Wout, float* __restrict__ wsum, float* __restrict__ lnb, int K) {
    const int n = blockIdx.x, lane = threadIdx.x;
    float s = 0.f, b = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float w = (float)W[(long)n * K + k];
        const half_t wp = (half_t)(w * (float)gamma[k]);
        Wout[(long)n * K + k] = wp;
        s += (float)wp;
        b = fmaf((float)beta[k], w, b);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        b += __shfl_xor(b, o, 64);
    }
    if (lane == 0) {
        wsum[n] = s;
        lnb[n] = b + (bias ? (float)bias[n] : 0.f);
    }
}

__global__ void count_not_dirac_kernel(const half_t* __restrict__ w, int Co, int Ci, int k, unsigned* cnt) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)Co * Ci * k) return;
    int t = (int)(i % k), ci = (int)((i / k) % Ci), co = (int)(i / ((long)k * Ci));
    float expect = (co == ci && t == k / 2) ? 1.f : 0.f;
    if ((float)w[i] != expect) atomicAdd(cnt, 1u);
}
__global__ void count_nonzero_kernel(const half_t* __restrict__ w, long n, unsigned* cnt) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (float)w[i] != 0.f) atomicAdd(cnt, 1u);
}

// per-unit variants (slot i of a counter array): which temporal layers are trained?
__global__ void count_not_dirac_slot_kernel(const half_t* __restrict__ w, int Co, int Ci, int k, unsigned* cnt) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)Co * Ci * k) return;
    int t = (int)(i % k), ci = (int)((i / k) % Ci), co = (int)(i / ((long)k * Ci));
    float expect = (co == ci && t == k / 2) ? 1.f : 0.f;
    if ((float)w[i] != expect) atomicAdd(cnt, 1u);
}
// Conv1d weight [C][C][3] over frames -> a 3x3 conv weight [C][ky][kx][CP] on the geometry (rows = frames, columns = pixels) whose
// side columns (kx != 1) are zero: the temporal conv runs through the ordinary implicit-GEMM conv kernels
__global__ void embed_temporal_weight_kernel(const half_t* __restrict__ in, half_t* __restrict__ out, int C) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)C * 9 * C) return;
    const int ci = (int)(i % C), tap = (int)((i / C) % 9), co = (int)(i / ((long)C * 9));
    const int ky = tap / 3, kx = tap % 3;
    out[i] = kx == 1 ? in[((long)co * C + ci) * 3 + ky] : (half_t)0.f;
}
// trained temporal conv on few channels (conv_out: 4): y[b,f,p,co] = bias[co] + sum_t sum_ci W[co][ci][t] x[b,f+t-1,p,ci]
__global__ void temporal_conv_small_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, const half_t* __restrict__ w,
                                           const half_t* __restrict__ bias, int B, int F, long HW, int C) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * F * HW * C) return;
    const int co = (int)(i % C);
    const long row = i / C;
    const int f = (int)((row / HW) % F);
    float acc = (float)bias[co];
    for (int t = 0; t < 3; ++t) {
        const int ff = f + t - 1;
        if (ff < 0 || ff >= F) continue;
        const half_t* xr = x + (row + (long)(t - 1) * HW) * C;
        for (int ci = 0; ci < C; ++ci) acc += (float)w[((long)co * C + ci) * 3 + t] * (float)xr[ci];
    }
    y[i] = (half_t)acc;
}
// trained temporal attention (attention.py:336-346): every pixel attends over the F frames of its clip.  One thread per
// (branch, pixel, head, query frame); q | k | v rows [(b f) n, 3C]; F <= 32.  A fine-tuned-checkpoint path: correctness first.
template <int D8>
__global__ __launch_bounds__(256) void temporal_attn_kernel(const half_t* __restrict__ qkv, half_t* __restrict__ out, int B, int F, int N, int C,
                                                            int heads, float scale) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * N * heads * F) return;
    const int n = (int)(i % N);
    const int f = (int)((i / N) % F);
    const int h = (int)((i / ((long)N * F)) % heads);
    const int b = (int)(i / ((long)N * F * heads));
    const long ld = 3L * C;
    const half_t* base = qkv + ((long)b * F * N + n) * ld + h * (D8 * 8);      // frame 0 of this pixel
    const long fs = (long)N * ld;                                             // frame stride
    h8 q[D8];
#pragma unroll
    for (int c = 0; c < D8; ++c) q[c] = *reinterpret_cast<const h8*>(base + f * fs + c * 8);
    float p[32];
    float mx = -INFINITY;
#pragma unroll
    for (int g = 0; g < 32; ++g) {
        p[g] = -INFINITY;
        if (g < F) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < D8; ++c) {
                const h8 k8 = *reinterpret_cast<const h8*>(base + g * fs + C + c * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) s += (float)q[c][e] * (float)k8[e];
            }
            p[g] = s * scale;
            mx = fmaxf(mx, p[g]);
        }
    }
    float l = 0.f;
#pragma unroll
    for (int g = 0; g < 32; ++g) {
        p[g] = g < F ? __expf(p[g] - mx) : 0.f;
        l += p[g];
    }
    const float inv = 1.f / l;
    half_t* op = out + (((long)b * F + f) * N + n) * C + h * (D8 * 8);
#pragma unroll 1
    for (int c = 0; c < D8; ++c) {
        float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 32; ++g) {
            if (g < F) {
                const h8 v8 = *reinterpret_cast<const h8*>(base + g * fs + 2 * C + c * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] += p[g] * (float)v8[e];
            }
        }
        h8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = (half_t)(o[e] * inv);
        *reinterpret_cast<h8*>(op + c * 8) = r;
    }
}

inline unsigned nb(long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

// ------------------------------------------------------------------------------------------ arena
void* Arena::alloc(size_t bytes) {
    bytes = (bytes + 255) & ~size_t(255);
    for (size_t i = 0; i < blocks.size(); ++i) {
        if (blocks[i].free && blocks[i].size >= bytes) {
            if (blocks[i].size > bytes) {
                Block rest{blocks[i].off + bytes, blocks[i].size - bytes, true};
                blocks[i].size = bytes;
                blocks.insert(blocks.begin() + i + 1, rest);
            }
            blocks[i].free = false;
            size_t used = blocks[i].off + bytes;
            if (used > high_water) high_water = used;
            return base + blocks[i].off;
        }
    }
    return nullptr;
}
void Arena::release(void* p) {
    if (!p) return;
    size_t off = (char*)p - base;
    for (size_t i = 0; i < blocks.size(); ++i) {
        if (blocks[i].off == off && !blocks[i].free) {
            blocks[i].free = true;
            if (i + 1 < blocks.size() && blocks[i + 1].free) {
                blocks[i].size += blocks[i + 1].size;
                blocks.erase(blocks.begin() + i + 1);
            }
            if (i > 0 && blocks[i - 1].free) {
                blocks[i - 1].size += blocks[i].size;
                blocks.erase(blocks.begin() + i);
            }
            return;
        }
    }
}
void Arena::reset() {
    blocks.clear();
    blocks.push_back(Block{0, size, true});
}

// ------------------------------------------------------------------------------------------ handle
UNet::~UNet() {
    for (auto& kv : weights) (void)hipFree(kv.second.ptr);
    for (auto& kv : derived) (void)hipFree(kv.second.ptr);
    for (auto& kv : idx_tables) (void)hipFree(kv.second);
    if (arena.base) (void)hipFree(arena.base);
    if (d_counter) (void)hipFree(d_counter);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (xstream) (void)hipStreamDestroy(xstream);
    if (emu_flag) (void)hipFree(emu_flag);
}

int UNet::comm_streams() {
    if (xstream) return UV_OK;
    int lo = 0, hi = 0;                         // (numerically lowest = greatest priority): the exchange's few kernels go ahead of the queued compute
    UV_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    static const int prio_env = getenv("UNIVST_XSTREAM_PRIO") ? atoi(getenv("UNIVST_XSTREAM_PRIO")) : 1;      // 0: default priority (A/B aid)
    UV_HIP(hipStreamCreateWithPriority(&xstream, hipStreamNonBlocking, prio_env ? hi : lo));
    UV_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    UV_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
    return UV_OK;
}

int UNet::load_tensor(const char* key, const void* dev_ptr, int dtype, const int64_t* shape, int ndim, hipStream_t s) {
    UV_REQUIRE(key && dev_ptr && ndim >= 1 && ndim <= 5, "load_tensor: bad arguments");
    long n = 1;
    WTensor t;
    for (int i = 0; i < ndim; ++i) {
        n *= shape[i];
        t.shape.push_back(shape[i]);
    }
    UV_REQUIRE(n > 0, "load_tensor(%s): empty tensor", key);
    UV_HIP(hipMalloc(&t.ptr, n * sizeof(half_t)));
    if (dtype == 0) {
        UV_HIP(hipMemcpyAsync(t.ptr, dev_ptr, n * sizeof(half_t), hipMemcpyDeviceToDevice, s));
    } else if (dtype == 1) {
        hipLaunchKernelGGL(convert_f32_f16_kernel, dim3(nb(n)), dim3(256), 0, s, (const float*)dev_ptr, t.ptr, n);
        UV_LAUNCH_CHECK();
    } else {
        (void)hipFree(t.ptr);
        uv_set_error("load_tensor(%s): dtype %d unsupported", key, dtype);
        return UV_ERR_ARG;
    }
    auto it = weights.find(key);
    if (it != weights.end()) {
        UV_HIP(hipStreamSynchronize(s));
        (void)hipFree(it->second.ptr);
        weights.erase(it);
    }
    weights[key] = t;
    finalized = false;
    return UV_OK;
}

const WTensor* UNet::find(const std::string& k) const {
    auto it = weights.find(k);
    if (it != weights.end()) return &it->second;
    auto jt = derived.find(k);
    return jt == derived.end() ? nullptr : &jt->second;
}

half_t* UNet::W(const std::string& k) {
    const WTensor* t = find(k);
    if (!t) {
        if (missing.empty()) missing = k;
        return nullptr;
    }
    return t->ptr;
}

int UNet::derive_alloc(const std::string& k, std::vector<long> shape, half_t** out) {
    auto it = derived.find(k);
    if (it != derived.end()) {
        (void)hipFree(it->second.ptr);
        derived.erase(it);
    }
    long n = 1;
    for (long v : shape) n *= v;
    WTensor t;
    t.shape = shape;
    UV_HIP(hipMalloc(&t.ptr, n * sizeof(half_t)));
    derived[k] = t;
    *out = t.ptr;
    return UV_OK;
}

int UNet::finalize(hipStream_t s) {
    std::vector<std::string> keys;
    for (auto& kv : weights) keys.push_back(kv.first);
    // one counter per temporal UNIT (a conv's temporal weight + bias, a block's attn_temporal.to_out weight): non-zero = trained
    std::vector<std::pair<std::string, int>> tunits;          // (prefix, 0 conv / 1 attention)
    std::unordered_map<std::string, int> tslot;
    for (const std::string& k : keys) {
        size_t pos;
        if ((pos = k.find(".conv_temporal.")) != std::string::npos) {
            const std::string pre = k.substr(0, pos);
            if (!tslot.count(pre)) { tslot[pre] = (int)tunits.size(); tunits.push_back({pre, 0}); }
        } else if ((pos = k.find(".attn_temporal.to_out.0.weight")) != std::string::npos) {
            const std::string pre = k.substr(0, pos);
            if (!tslot.count(pre + "#a")) { tslot[pre + "#a"] = (int)tunits.size(); tunits.push_back({pre, 1}); }
        }
    }
    const size_t ncnt = tunits.size() + 1;
    if (d_counter) (void)hipFree(d_counter);
    d_counter = nullptr;
    UV_HIP(hipMalloc(&d_counter, ncnt * sizeof(unsigned)));
    UV_HIP(hipMemsetAsync(d_counter, 0, ncnt * sizeof(unsigned), s));
    temporal_conv_active.clear();
    temporal_attn_active.clear();
    auto ends = [](const std::string& a, const char* suf) {
        size_t n = strlen(suf);
        return a.size() >= n && a.compare(a.size() - n, n, suf) == 0;
    };
    auto level_of = [](const std::string& a) {      // resolution level (0 = finest) of a state-dict key
        int i = 0;
        if (sscanf(a.c_str(), "down_blocks.%d.", &i) == 1) return i;
        if (sscanf(a.c_str(), "up_blocks.%d.", &i) == 1) return 3 - i;
        return 3;                                    // mid_block
    };
    for (const std::string& k : keys) {
        const WTensor& t = weights[k];
        if (k.find("conv_temporal.weight") != std::string::npos) {
            UV_REQUIRE(t.shape.size() == 3 && t.shape[0] == t.shape[1] && t.shape[2] == 3, "%s: expected a [C,C,3] Conv1d weight", k.c_str());
            long n = t.shape[0] * t.shape[1] * t.shape[2];
            hipLaunchKernelGGL(count_not_dirac_slot_kernel, dim3(nb(n)), dim3(256), 0, s, t.ptr, (int)t.shape[0], (int)t.shape[1],
                               (int)t.shape[2], d_counter + tslot[k.substr(0, k.find(".conv_temporal."))]);
        } else if (k.find("conv_temporal.bias") != std::string::npos) {
            long n = t.shape[0];
            hipLaunchKernelGGL(count_nonzero_kernel, dim3(nb(n)), dim3(256), 0, s, t.ptr, n, d_counter + tslot[k.substr(0, k.find(".conv_temporal."))]);
        } else if (ends(k, "attn_temporal.to_out.0.weight")) {
            long n = 1;
            for (long v : t.shape) n *= v;
            hipLaunchKernelGGL(count_nonzero_kernel, dim3(nb(n)), dim3(256), 0, s, t.ptr, n,
                               d_counter + tslot[k.substr(0, k.find(".attn_temporal.to_out.0.weight")) + "#a"]);
        } else if (t.shape.size() == 4 && k.find("_temporal") == std::string::npos) {
            // conv weights -> [Co][taps][CiP]
            int Co = (int)t.shape[0], Ci = (int)t.shape[1], taps = (int)(t.shape[2] * t.shape[3]);
            UV_REQUIRE(taps == 1 || taps == 9, "%s: only 1x1 / 3x3 convs are supported", k.c_str());
            int CiP = (Ci + 7) / 8 * 8;
            half_t* d;
            int rc = derive_alloc(k + "#nhwc", {Co, taps, CiP}, &d);
            if (rc) return rc;
            long n = (long)Co * taps * CiP;
            hipLaunchKernelGGL(permute_conv_weight_kernel, dim3(nb(n)), dim3(256), 0, s, t.ptr, d, Co, Ci, taps, CiP);
            if (taps == 9 && Ci % 64 == 0) {      // tap-inner copy (GemmParams::korder = 1)
                half_t* d2;
                rc = derive_alloc(k + "#ti", {Co, Ci / 64, 9, 64}, &d2);
                if (rc) return rc;
                hipLaunchKernelGGL(permute_conv_weight_ti_kernel, dim3(nb((long)Co * Ci * 9)), dim3(256), 0, s, t.ptr, d2, Co, Ci);
            }
            if (taps == 9 && Ci % 64 == 0 && Co % 320 == 0 && k.find("downsamplers") == std::string::npos) {      // LDS-patch copy (GemmParams::W32; stride-1 convs)
                half_t* d3;
                rc = derive_alloc(k + "#t32", {Co, Ci / 32, 9, 32}, &d3);
                if (rc) return rc;
                hipLaunchKernelGGL(permute_conv_weight_t32_kernel, dim3(nb((long)Co * Ci * 9)), dim3(256), 0, s, t.ptr, d3, Co, Ci);
            }
        } else if (ends(k, ".attn1.to_q.weight")) {
            std::string p = k.substr(0, k.size() - strlen("to_q.weight"));
            const WTensor *tk = find(p + "to_k.weight"), *tv = find(p + "to_v.weight");
            UV_REQUIRE(tk && tv, "%s: to_k / to_v missing", p.c_str());
            long C = t.shape[0], K = t.shape[1];
            half_t* d;
            int rc = derive_alloc(p + "qkv#fused", {3 * C, K}, &d);
            if (rc) return rc;
            UV_HIP(hipMemcpyAsync(d, t.ptr, C * K * 2, hipMemcpyDeviceToDevice, s));
            UV_HIP(hipMemcpyAsync(d + C * K, tk->ptr, C * K * 2, hipMemcpyDeviceToDevice, s));
            UV_HIP(hipMemcpyAsync(d + 2 * C * K, tv->ptr, C * K * 2, hipMemcpyDeviceToDevice, s));
            const int hd = (int)C / cfg.attention_heads[level_of(k)];
            if (hd == 40 || hd == 64 || hd == 80 || hd == 160) {         // second copy whose Q rows carry log2(e)/sqrt(d) (AttnParams::q_prescaled): the software-
                half_t* d2;                     // pipelined kernels (head_dim 40: SD-v1.5; 64: the SD-v2.1 layout) take the scale from the weights
                rc = derive_alloc(p + "qkv#fused#qs", {3 * C, K}, &d2);
                if (rc) return rc;
                hipLaunchKernelGGL(scale_f16_kernel, dim3(nb(C * K)), dim3(256), 0, s, t.ptr, d2, C * K, 1.4426950408889634f / sqrtf((float)hd));
                UV_HIP(hipMemcpyAsync(d2 + C * K, tk->ptr, C * K * 2, hipMemcpyDeviceToDevice, s));
                UV_HIP(hipMemcpyAsync(d2 + 2 * C * K, tv->ptr, C * K * 2, hipMemcpyDeviceToDevice, s));
            }
        } else if (ends(k, ".attn2.to_q.weight")) {
            const long C = t.shape[0], K = t.shape[1];
            const int hd = (int)C / cfg.attention_heads[level_of(k)];
            if (hd == 40) {
                half_t* d2;
                int rc = derive_alloc(k + "#qs", {C, K}, &d2);
                if (rc) return rc;
                hipLaunchKernelGGL(scale_f16_kernel, dim3(nb(C * K)), dim3(256), 0, s, t.ptr, d2, C * K, 1.4426950408889634f / sqrtf((float)hd));
            }
        } else if (ends(k, ".attn2.to_k.weight")) {
            std::string p = k.substr(0, k.size() - strlen("to_k.weight"));
            const WTensor* tv = find(p + "to_v.weight");
            UV_REQUIRE(tv, "%s: to_v missing", p.c_str());
            long C = t.shape[0], K = t.shape[1];
            half_t* d;
            int rc = derive_alloc(p + "kv#fused", {2 * C, K}, &d);
            if (rc) return rc;
            UV_HIP(hipMemcpyAsync(d, t.ptr, C * K * 2, hipMemcpyDeviceToDevice, s));
            UV_HIP(hipMemcpyAsync(d + C * K, tv->ptr, C * K * 2, hipMemcpyDeviceToDevice, s));
        } else if (ends(k, ".ff.net.0.proj.weight") || ends(k, ".ff.net.0.proj.bias")) {
            int rows = (int)t.shape[0], cols = t.shape.size() > 1 ? (int)t.shape[1] : 1;
            UV_REQUIRE(rows % 32 == 0, "%s: GEGLU width %d must be a multiple of 32", k.c_str(), rows);
            half_t* d;
            int rc = derive_alloc(k + "#geglu", {rows, cols}, &d);
            if (rc) return rc;
            hipLaunchKernelGGL(geglu_interleave_kernel, dim3(nb((long)rows * cols)), dim3(256), 0, s, t.ptr, d, rows, cols);
            // K = 320 (the 64x64 level): a second copy in the row order of the X-resident kernel (gemm.hip geglu_xres_kernel)
            int kdim = cols;
            if (!ends(k, ".weight")) {      // the bias: the reduction length is its weight's (a checkpoint that carries the bias without the weight is refused)
                const WTensor* wt = find(k.substr(0, k.size() - strlen("bias")) + "weight");
                UV_REQUIRE(wt && wt->shape.size() == 2, "%s: the projection's weight is missing", k.c_str());
                kdim = (int)wt->shape[1];
            }
            if (uv_geglu_xres_ok(rows, kdim)) {
                half_t* dx;
                rc = derive_alloc(k + "#xres", {rows, cols}, &dx);
                if (rc) return rc;
                rc = uv_launch_geglu_xres_permute(t.ptr, dx, rows, cols, s);
                if (rc) return rc;
            }
        }
    }
    // LayerNorm-folded copies of the three linears that follow norm1 / norm2 / norm3 of every transformer block
    for (const std::string& k : keys) {
        if (!ends(k, ".transformer_blocks.0.norm1.weight")) continue;
        const std::string b = k.substr(0, k.size() - strlen(".norm1.weight"));
        struct FoldJob { const char* norm; std::string w; std::string bias; };
        std::vector<FoldJob> jobs = {
            {".norm1", b + (find(b + ".attn1.qkv#fused#qs") ? ".attn1.qkv#fused#qs" : ".attn1.qkv#fused"), ""},
            {".norm2", b + (find(b + ".attn2.to_q.weight#qs") ? ".attn2.to_q.weight#qs" : ".attn2.to_q.weight"), ""},
            {".norm3", b + ".ff.net.0.proj.weight#geglu", b + ".ff.net.0.proj.bias#geglu"}};
        if (find(b + ".ff.net.0.proj.weight#xres")) jobs.push_back({".norm3", b + ".ff.net.0.proj.weight#xres", b + ".ff.net.0.proj.bias#xres"});
        for (auto& j : jobs) {
            const WTensor *w = find(j.w), *gm = find(b + j.norm + ".weight"), *bt = find(b + j.norm + ".bias");
            const WTensor* bi = j.bias.empty() ? nullptr : find(j.bias);
            UV_REQUIRE(w && gm && bt && (j.bias.empty() || bi) && w->shape.size() == 2 && gm->shape[0] == w->shape[1],
                       "%s: LayerNorm / linear pair incomplete", (b + j.norm).c_str());
            const long N = w->shape[0], K = w->shape[1];
            half_t *wo, *ws, *lb;
            int rc = derive_alloc(j.w + "#ln", {N, K}, &wo);
            if (!rc) rc = derive_alloc(j.w + "#ln.wsum", {2 * N}, &ws);      // fp32 [N]
            if (!rc) rc = derive_alloc(j.w + "#ln.bias", {2 * N}, &lb);      // fp32 [N]
            if (rc) return rc;
            hipLaunchKernelGGL(ln_fold_weight_kernel, dim3((unsigned)N), dim3(64), 0, s, w->ptr, gm->ptr, bt->ptr, bi ? bi->ptr : nullptr, wo,
                               (float*)ws, (float*)lb, (int)K);
        }
    }
    // the fused text cross-attention (fused.hip, the 64x64 level of SD-v1.x): to_q (plain and LayerNorm-folded) and to_out in MFMA operand order
    for (const std::string& k : keys) {
        if (!ends(k, ".attn2.to_out.0.weight")) continue;
        const std::string b = k.substr(0, k.size() - strlen(".attn2.to_out.0.weight"));
        const WTensor& t = weights[k];
        if (t.shape.size() != 2 || t.shape[0] != t.shape[1] || !uv_attn2_fused_ok((int)t.shape[0], cfg.attention_heads[level_of(k)], 64, 77)) continue;
        const std::string wq = b + (find(b + ".attn2.to_q.weight#qs") ? ".attn2.to_q.weight#qs" : ".attn2.to_q.weight");
        for (const std::string& src : {k, wq, wq + "#ln", b + ".attn1.to_out.0.weight"}) {
            const WTensor* w = find(src);
            if (!w) continue;
            half_t* d;
            int rc = derive_alloc(src + "#frag", {w->shape[0], w->shape[1]}, &d);
            if (rc) return rc;
            rc = uv_launch_frag_pack(w->ptr, d, (int)w->shape[0], (int)w->shape[1], s);
            if (rc) return rc;
        }
    }
    {   // all resnets' time_emb_proj stacked into one [sum Cout, 4*C0] matrix: one projection launch per forward instead of 22
        std::vector<std::string> tk;
        for (const std::string& k : keys)
            if (ends(k, ".time_emb_proj.weight")) tk.push_back(k);
        std::sort(tk.begin(), tk.end());
        temb_off.clear();
        temb_total = 0;
        for (const std::string& k : tk) {
            temb_off[k.substr(0, k.size() - strlen(".time_emb_proj.weight"))] = temb_total;
            temb_total += weights[k].shape[0];
        }
        if (!tk.empty()) {
            const long K = weights[tk[0]].shape[1];
            half_t *wa, *ba;
            int rc = derive_alloc("time_emb_proj#all.weight", {temb_total, K}, &wa);
            if (rc) return rc;
            rc = derive_alloc("time_emb_proj#all.bias", {temb_total}, &ba);
            if (rc) return rc;
            for (const std::string& k : tk) {
                const std::string pre = k.substr(0, k.size() - strlen(".time_emb_proj.weight"));
                const WTensor& wt = weights[k];
                const WTensor* bt = find(pre + ".time_emb_proj.bias");
                UV_REQUIRE(bt && wt.shape[1] == K, "%s: time_emb_proj bias missing or width mismatch", pre.c_str());
                UV_HIP(hipMemcpyAsync(wa + temb_off[pre] * K, wt.ptr, wt.shape[0] * K * sizeof(half_t), hipMemcpyDeviceToDevice, s));
                UV_HIP(hipMemcpyAsync(ba + temb_off[pre], bt->ptr, wt.shape[0] * sizeof(half_t), hipMemcpyDeviceToDevice, s));
            }
        }
    }
    UV_LAUNCH_CHECK();
    std::vector<unsigned> hc(ncnt, 0);
    UV_HIP(hipMemcpyAsync(hc.data(), d_counter, ncnt * sizeof(unsigned), hipMemcpyDeviceToHost, s));
    UV_HIP(hipStreamSynchronize(s));
    // trained temporal layers: derived layouts for the units that need them (resnet.py:70-80, attention.py:336-346)
    for (size_t i = 0; i < tunits.size(); ++i) {
        if (!hc[i]) continue;
        const std::string& pre = tunits[i].first;
        if (tunits[i].second == 0) {
            const WTensor* w = find(pre + ".conv_temporal.weight");
            UV_REQUIRE(w && find(pre + ".conv_temporal.bias"), "%s: conv_temporal weight / bias incomplete", pre.c_str());
            const int C = (int)w->shape[0];
            if (C % 8 == 0) {
                half_t* d;
                int rc = derive_alloc(pre + ".conv_temporal.weight#t3x3", {C, 9, C}, &d);
                if (rc) return rc;
                hipLaunchKernelGGL(embed_temporal_weight_kernel, dim3(nb((long)C * 9 * C)), dim3(256), 0, s, w->ptr, d, C);
            }
            temporal_conv_active[pre] = 1;
        } else {
            const WTensor *tq = find(pre + ".attn_temporal.to_q.weight"), *tk = find(pre + ".attn_temporal.to_k.weight"),
                          *tv = find(pre + ".attn_temporal.to_v.weight");
            UV_REQUIRE(tq && tk && tv && find(pre + ".norm_temporal.weight") && find(pre + ".norm_temporal.bias") &&
                       find(pre + ".attn_temporal.to_out.0.bias"), "%s: attn_temporal / norm_temporal parameters incomplete", pre.c_str());
            const long C = tq->shape[0], K = tq->shape[1];
            half_t* d;
            int rc = derive_alloc(pre + ".attn_temporal.qkv#fused", {3 * C, K}, &d);
            if (rc) return rc;
            UV_HIP(hipMemcpyAsync(d, tq->ptr, C * K * 2, hipMemcpyDeviceToDevice, s));
            UV_HIP(hipMemcpyAsync(d + C * K, tk->ptr, C * K * 2, hipMemcpyDeviceToDevice, s));
            UV_HIP(hipMemcpyAsync(d + 2 * C * K, tv->ptr, C * K * 2, hipMemcpyDeviceToDevice, s));
            temporal_attn_active[pre] = 1;
        }
    }
    UV_LAUNCH_CHECK();
    UV_HIP(hipStreamSynchronize(s));
    finalized = true;
    return UV_OK;
}

int UNet::reserve(int B, int F, int H, int Wd) {
    const long rows0 = (long)B * F * H * Wd;
    // activations (28 level-0-sized tensors at the high-water mark of the graph) + the small workspaces + split-K partials + the pool of the
    // producers' GroupNorm statistics (Fwd::gst_pool: rows0 / 16 fragments x 32 sub-group slots x 64 tensors x (sum, sumsq) fp32 = 1 KB per row)
    size_t need = (size_t)rows0 * cfg.block_out_channels[0] * 2 * 28 + (64u << 20) + UV_SPLITK_WS_BYTES + (gn_producer ? (size_t)rows0 * 1024 : 0);
    if (arena.size < need) {
        UV_HIP(hipDeviceSynchronize());
        if (arena.base) UV_HIP(hipFree(arena.base));
        arena.base = nullptr;
        UV_HIP(hipMalloc((void**)&arena.base, need));
        arena.size = need;
    }
    arena.reset();
    // K/V source tables for this (B,F)
    long key = ((long)B << 20) | F | ((long)rank << 40) | ((long)world << 50);
    if (!idx_tables.count(key)) {
        std::vector<int> t;
        // frame shard: ranks > 0 find the previous frame of their first local frame and the clip's frame 0 in the
        // extra K/V frames appended after the B*F local ones ([B prev | B first], filled by Fwd::kv_exchange)
        const bool ext = world > 1 && rank > 0;
        auto prev = [&](int b, int f) { return f > 0 ? b * F + f - 1 : (ext ? B * F + b : b * F); };
        auto first = [&](int b) { return ext ? B * F + B + b : b * F; };
        // Duplicate sources are merged: at f = 0 the reference's key set is {0,0,0} (stock) / {0,0} (PnP), at f = 1 it is
        // {0,1,0} / {0,0}.  Softmax over a key set that contains frame X m times equals reading X once with log2(m) added
        // to its scores, so the table keeps unique sources + (count, log2 multiplicity) — exact, 6 % fewer key tiles.
        std::vector<int> cnt;
        std::vector<float> lw;
        auto emit = [&](std::vector<int> srcs, int width) {
            std::vector<int> uniq;
            std::vector<int> mult;
            for (int sidx : srcs) {
                size_t j = 0;
                for (; j < uniq.size(); ++j)
                    if (uniq[j] == sidx) break;
                if (j == uniq.size()) { uniq.push_back(sidx); mult.push_back(1); }
                else mult[j]++;
            }
            cnt.push_back((int)uniq.size());
            for (int j = 0; j < width; ++j) {
                t.push_back(j < (int)uniq.size() ? uniq[j] : uniq[0]);
                lw.push_back(j < (int)uniq.size() ? log2f((float)mult[j]) : 0.f);
            }
        };
        // Round 6, ranks > 0 of a frame shard: the key set of every query is split into the frames this rank HOLDS (phase 1, runs while the halo frames
        // are on the wire) and the two halo frames (phase 2, continues from phase 1's softmax state): the primary tables carry the local sources — a
        // frame may have none: PnP at f = 0 — and a second block [idx stock 3BF | idx pnp 2BF | cnt stock BF | cnt pnp BF] behind them the halo ones
        // (never duplicates: prev and first are different row blocks even where they are the same global frame).
        auto emit_split = [&](std::vector<int> local, std::vector<int> halo, int width, std::vector<int>& t2, std::vector<int>& cnt2, int self) {
            cnt.push_back((int)local.size());
            for (int j = 0; j < width; ++j) {
                t.push_back(j < (int)local.size() ? local[j] : self);
                lw.push_back(0.f);
            }
            cnt2.push_back((int)halo.size());
            for (int j = 0; j < width; ++j) t2.push_back(j < (int)halo.size() ? halo[j] : self);
        };
        std::vector<int> t2s, t2p, c2s, c2p;
        for (int b = 0; b < B; ++b)
            for (int f = 0; f < F; ++f) {                                             // stock: [-1, 0, 'first'] (attention.py:356)
                if (!ext) emit({prev(b, f), b * F + f, first(b)}, 3);
                else if (f == 0) emit_split({b * F + f}, {prev(b, f), first(b)}, 3, t2s, c2s, b * F + f);
                else emit_split({prev(b, f), b * F + f}, {first(b)}, 3, t2s, c2s, b * F + f);
            }
        for (int b = 0; b < B; ++b)
            for (int f = 0; f < F; ++f) {                                             // PnP: [-1, 'first'] (pnp_utils.py:25)
                if (!ext) emit({prev(b, f), first(b)}, 2);
                else if (f == 0) emit_split({}, {prev(b, f), first(b)}, 2, t2p, c2p, b * F + f);
                else emit_split({prev(b, f)}, {first(b)}, 2, t2p, c2p, b * F + f);
            }
        for (int b = 0; b < B; ++b)
            for (int f = 0; f < F; ++f) t.push_back(b);   // text: one [77, C] block per branch
        // layout of the device table: [idx stock 3BF | idx pnp 2BF | idx text BF | cnt stock BF | cnt pnp BF | logw stock 3BF | logw pnp 2BF]
        // (+ on ranks > 0 of a shard: [idx stock2 3BF | idx pnp2 2BF | cnt stock2 BF | cnt pnp2 BF])
        for (int v : cnt) t.push_back(v);
        for (float v : lw) {
            int bits;
            memcpy(&bits, &v, 4);
            t.push_back(bits);
        }
        for (auto* v : {&t2s, &t2p, &c2s, &c2p})
            for (int x : *v) t.push_back(x);
        int* d;
        UV_HIP(hipMalloc(&d, t.size() * sizeof(int)));
        UV_HIP(hipMemcpy(d, t.data(), t.size() * sizeof(int), hipMemcpyHostToDevice));
        idx_tables[key] = d;
    }
    return UV_OK;
}

// ------------------------------------------------------------------------------------------ forward
#define RUN(x)                \
    do {                      \
        int _rc = (x);        \
        if (_rc) return _rc;  \
    } while (0)

struct Fwd {
    UNet& u;
    hipStream_t s;
    int B, F, text_len;
    const univst_pnp_t* pnp;
    half_t* emb = nullptr;     // [B, 4*C0]
    const half_t* text = nullptr;
    const int *idx_stock = nullptr, *idx_pnp = nullptr, *idx_text = nullptr, *cnt_stock = nullptr, *cnt_pnp = nullptr;
    const float *lw_stock = nullptr, *lw_pnp = nullptr;
    const int *idx2_stock = nullptr, *idx2_pnp = nullptr, *cnt2_stock = nullptr, *cnt2_pnp = nullptr;   // halo phase (ranks > 0 of a frame shard)
    float* gn_ws = nullptr;
    float* sk_ws = nullptr;        // split-K fp32 partials (UV_SPLITK_WS_BYTES)
    half_t* temb_all = nullptr;    // [B, temb_total]: every resnet's time_emb_proj(SiLU(emb)), one launch per forward
    float* ad_ws = nullptr;
    float* gst_pool = nullptr;     // bump region for the producers' GroupNorm statistics (never reused inside a forward)
    size_t gst_left = 0;           // floats

    float* gst_take(long rows, int C) {   // [C/10][rows/16][2] floats, or null (pool exhausted / switched off: the consumer runs its own pass)
        const size_t n = (size_t)(rows / 16) * (C / 10) * 2;
        if (!u.gn_producer || !gst_pool || rows % 16 != 0 || C % 10 != 0 || n > gst_left) return nullptr;
        float* p = gst_pool;
        gst_pool += n;
        gst_left -= n;
        return p;
    }
    void gst_give_back(long rows, int C) {
        const size_t n = (size_t)(rows / 16) * (C / 10) * 2;
        gst_pool -= n;
        gst_left += n;
    }

    half_t* alloc(long elems) {
        half_t* p = (half_t*)u.arena.alloc((size_t)elems * sizeof(half_t));
        if (!p) uv_set_error("activation arena exhausted (%zu bytes); call univst_unet_reserve with the right geometry", u.arena.size);
        return p;
    }
    void free(void* p) { u.arena.release(p); }
    half_t* W(const std::string& k) { return u.W(k); }

    int groupnorm(const Act& a, const Act* b, int rows_per_stat, float eps, const std::string& p, int silu, half_t* out,
                  bool cross_frame = false, const UvGnFold* fold = nullptr) {
        UvGnComm gc;
        const bool sharded = cross_frame && u.world > 1;
        if (sharded) {             // 5-D GroupNorm statistics span all frames of a branch => sum the partials over ranks
            gc.world = u.world;
            gc.red = (float*)u.comm_ws;
            gc.byte_off = 0;
            gc.allreduce = u.allreduce;
            gc.user = u.comm_user;
        }
        return uv_launch_groupnorm(a.p, b ? b->p : nullptr, a.C, b ? b->C : 0, a.rows(), rows_per_stat, u.cfg.norm_num_groups,
                                   eps, W(p + ".weight"), W(p + ".bias"), silu, out, gn_ws, s, sharded ? &gc : nullptr,
                                   rows_per_stat % 16 == 0 ? a.gst : nullptr, b ? b->gst : nullptr, fold);
    }
    // want_gst: let the epilogue leave the GroupNorm statistics of the output (Act::gst) for a following GroupNorm
    int conv(const Act& a, const Act* b, const std::string& p, int Cout, int taps, int stride, int up, const half_t* rowbias,
             const half_t* R, Act* out, long ldrb = 0, bool want_gst = false) {
        GemmParams g;
        g.X = a.p;
        g.X2 = b ? b->p : nullptr;
        g.C1 = a.C;
        g.C2 = b ? b->C : 0;
        g.Hs = a.H;
        g.Ws = a.W;
        g.up = up;
        g.stride = stride;
        g.taps = taps;
        const int He = a.H << up, We = a.W << up;
        g.Ho = (taps == 9) ? (He + 2 - 3) / stride + 1 : He;
        g.Wo = (taps == 9) ? (We + 2 - 3) / stride + 1 : We;
        g.M = a.imgs * g.Ho * g.Wo;
        g.N = Cout;
        g.K = taps * (g.C1 + g.C2);
        g.korder = (taps == 9 && g.C1 % 64 == 0 && g.C2 % 64 == 0 && u.find(p + ".weight#ti")) ? 1 : 0;
        g.W = W(p + (g.korder ? ".weight#ti" : ".weight#nhwc"));
        if (taps == 9 && stride == 1 && u.find(p + ".weight#t32")) g.W32 = W(p + ".weight#t32");
        g.bias = W(p + ".bias");
        g.rowbias = rowbias;
        g.ldrb = ldrb;
        g.rows_per_rb = F * g.Ho * g.Wo;
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
        if (!g.W || !g.bias) return u.missing_error();
        g.partial = sk_ws;
        g.partial_bytes = UV_SPLITK_WS_BYTES;
        out->gst = nullptr;
        if (taps != 9 || !u.temporal_conv_active.count(p)) {
            int emitted = 0;
            if (want_gst && (g.gn_out = gst_take(out->rows(), Cout))) {
                g.gn_G = Cout / 10;           // sub-groups of 10 channels: every group width of the UNet (10 .. 80) is a multiple
                g.gn_gw = 10;
                g.gn_emitted = &emitted;
            }
            RUN(uv_launch_gemm(g, 1, s));
            if (g.gn_out) {
                if (emitted) out->gst = g.gn_out;
                else gst_give_back(out->rows(), Cout);
            }
            return UV_OK;
        }
        // TRAINED temporal conv (resnet.py:70-80): spatial conv (+ its bias) -> Conv1d over the frames of every pixel -> whatever the
        // caller wanted fused behind the PseudoConv3d (time-embedding row bias, residual).  The Conv1d runs as a 3x3 conv on the
        // geometry (image rows = frames, image columns = pixels) with zero side taps (embed_temporal_weight_kernel).
        half_t* mid = alloc(out->rows() * Cout);
        if (!mid) return UV_ERR_STATE;
        g.Y = mid;
        g.rowbias = nullptr;
        g.R = nullptr;
        RUN(uv_launch_gemm(g, 1, s));
        half_t *tw = W(p + ".conv_temporal.weight"), *tb = W(p + ".conv_temporal.bias");
        if (!tw || !tb) return u.missing_error();
        const long HWo = (long)g.Ho * g.Wo;
        if (Cout % 8 != 0) {
            UV_REQUIRE(!rowbias && !R, "%s: temporal conv on %d channels cannot carry a fused epilogue", p.c_str(), Cout);
            hipLaunchKernelGGL(temporal_conv_small_kernel, dim3(nb(out->rows() * Cout)), dim3(256), 0, s, mid, out->p, tw, tb, B, F, HWo, Cout);
            UV_LAUNCH_CHECK();
        } else {
            GemmParams t;
            t.X = mid;
            t.C1 = Cout;
            t.Hs = F;
            t.Ws = (int)HWo;
            t.taps = 9;
            t.Ho = F;
            t.Wo = (int)HWo;
            t.M = (int)out->rows();
            t.N = Cout;
            t.K = 9 * Cout;
            t.W = W(p + ".conv_temporal.weight#t3x3");
            t.bias = tb;
            t.rowbias = rowbias;
            t.ldrb = ldrb;
            t.rows_per_rb = (int)(F * HWo);
            t.R = R;
            t.ldr = Cout;
            t.Y = out->p;
            t.ldy = Cout;
            if (!t.W) return u.missing_error();
            t.partial = sk_ws;
            t.partial_bytes = UV_SPLITK_WS_BYTES;
            RUN(uv_launch_gemm(t, 1, s));
        }
        free(mid);
        return UV_OK;
    }
    // stats_out: emit the per-row (sum, sumsq) of Y for a following folded LayerNorm.  ln_in: fold LayerNorm(X) (statistics in ln_in,
    // ln_slots = K / 160 slots per row) into this linear: `wkey` then names the derived "#ln" weight and the bias comes with it.
    int linear(const half_t* X, long ldx, long M, int K, const std::string& wkey, const std::string& bkey, int N, half_t* Y,
               long ldy, const half_t* R = nullptr, long ldr = 0, const half_t* bias2 = nullptr, int geglu = 0, float* stats_out = nullptr,
               const float* ln_in = nullptr, const float** gst_out = nullptr, const half_t* wsets = nullptr, const float* bias32 = nullptr, int rows_per_set = 0) {
        GemmParams g;
        if (wsets) {                  // per-frame weight sets + fp32 bias (a GroupNorm folded into this linear: uv_launch_groupnorm's fold)
            g.w_rows_per_set = rows_per_set;
            g.bias32 = bias32;
        }
        int emitted = 0;
        if (gst_out) {
            *gst_out = nullptr;
            if ((g.gn_out = gst_take(M, N))) {
                g.gn_G = N / 10;
                g.gn_gw = 10;
                g.gn_emitted = &emitted;
            }
        }
        g.stats_out = stats_out;
        if (ln_in) {
            g.ln_stats = ln_in;
            g.ln_slots = K / 160;
            g.ln_eps = 1e-5f;
            g.ln_wsum = (const float*)W(wkey + ".wsum");
            g.ln_bias = (const float*)W(wkey + ".bias");
            if (!g.ln_wsum || !g.ln_bias) return u.missing_error();
        }
        g.X = X;
        g.ldx = ldx;
        g.M = (int)M;
        g.K = K;
        g.N = N;
        g.W = wsets ? wsets : W(wkey);
        g.bias = (bkey.empty() || ln_in || wsets) ? nullptr : W(bkey);
        g.Y = Y;
        g.ldy = ldy;
        g.R = R;
        g.ldr = ldr;
        g.bias2 = bias2;
        g.geglu = geglu;
        if (!g.W || (!bkey.empty() && !ln_in && !wsets && !g.bias)) return u.missing_error();
        g.partial = sk_ws;
        g.partial_bytes = UV_SPLITK_WS_BYTES;
        RUN(uv_launch_gemm(g, 0, s));
        if (g.gn_out) {
            if (emitted) *gst_out = g.gn_out;
            else gst_give_back(M, N);
        }
        return UV_OK;
    }

    // frame shard (SURVEY §8e coupling 2): every frame attends to {prev, (cur), first}; the previous frame of this rank's first frame lives on
    // rank-1 and frame 0 on rank 0.  Round 6: what travels is the block's HIDDEN rows of the boundary frame ([B, N, C] fp16 per pack: the input of
    // norm1 -> to_k | to_v, attention.py:311,375-377; half the bytes of the K|V pack) — the receiver projects (and, inside the PnP window, shifts:
    // the shift needs per-frame statistics only, pnp_utils.py:114-125) the two halo frames itself — and it travels on a FORKED stream as soon as
    // proj_in has written the rows, beside this rank's own q|k|v projection, AdaIN shift and the LOCAL phase of its attention:
    //   kv_post (after proj_in):  pack on s -> fork (one event) -> [xstream: multicast to the peers -> raise their flags]
    //   kv_join (after phase 1):  the wait kernel on s spins on this rank's own flags; the packs are then in the inbox slots of comm_ws
    // The streams meet through the flags only; xstream is joined back into s ONCE, at the end of the forward (UNet::forward).
    // A host-callback communicator (torch.distributed / gloo / the host-thread loopback of the tests) is driven from kv_join on s: same packs, serial.
    struct KvSlots {
        long o_send = 0, o_first = 0, o_prev = 0, o_rfirst = 0, nbytes = 0;
        bool emu = false;
    } kvs;
    int kv_post(const half_t* h, int C, int N) {
        kvs = KvSlots();
        kvs.nbytes = (long)B * N * C * sizeof(half_t);
        // host-callback communicator: 4 slots [send | first | recv prev | recv first].  Native (IPC) communicator: the two receive slots
        // are double-buffered by exchange parity (peers write them without an acknowledgement, comm.hip): 6 slots
        const int nslot = u.native_comm ? 6 : 4;
        const long slot = ((u.comm_ws_bytes - 65536) / nslot) & ~255L;
        UV_REQUIRE(kvs.nbytes <= slot, "kv_exchange: comm workspace too small (%ld B per slot, need %ld)", slot, kvs.nbytes);
        const long par = u.native_comm ? uv_comm_kv_parity(u.native_comm) : 0;
        kvs.o_send = 65536;
        kvs.o_first = kvs.o_send + slot;
        kvs.o_prev = kvs.o_first + slot * (1 + 2 * par);
        kvs.o_rfirst = kvs.o_prev + slot;
        if (u.rank < u.world - 1) RUN(uv_launch_rows_pack(h, C, 0, C, N, B, F, F - 1, (half_t*)(u.comm_ws + kvs.o_send), s));
        if (u.rank == 0) RUN(uv_launch_rows_pack(h, C, 0, C, N, B, F, 0, (half_t*)(u.comm_ws + kvs.o_first), s));
        kvs.emu = !u.native_comm && u.emu_wire_gbps > 0;
        if (!u.native_comm && !kvs.emu) return UV_OK;
        const bool sends = u.rank < u.world - 1 || kvs.emu;           // (the last rank posts nothing)
        hipStream_t x = s;
        if (u.kv_overlap && sends) {
            RUN(u.comm_streams());
            x = u.xstream;
            UV_HIP(hipEventRecord(u.ev_fork, s));
            UV_HIP(hipStreamWaitEvent(x, u.ev_fork, 0));
            u.x_dirty = true;
        }
        if (u.native_comm) {
            RUN(uv_comm_kv_post(u.native_comm, kvs.o_send, kvs.o_first, kvs.o_prev, kvs.o_rfirst, kvs.nbytes, x));
        } else {
            // the slowest transfer of this exchange on a node this box does not have: one pack per link (the first-frame pack reaches every rank over
            // its own link from rank 0, the halo pack over the link from rank - 1) except on rank 1, whose one link from rank 0 carries both.  The delay
            // kernel + a raise of this process's own flag word stand in for the peer's multicast + raise
            const double us = u.emu_wire_lat_us + (u.rank == 1 ? 2.0 : 1.0) * (double)kvs.nbytes / (u.emu_wire_gbps * 1e3);
            u.emu_wire_us += us;
            if (!u.emu_flag) {
                UV_HIP(hipMalloc((void**)&u.emu_flag, 64));
                UV_HIP(hipMemsetAsync(u.emu_flag, 0, 64, s));
                UV_HIP(hipStreamSynchronize(s));
            }
            RUN(uv_launch_delay_us(us, x));
            RUN(uv_comm_launch_raise(u.emu_flag, ++u.emu_epoch, x));
        }
        return UV_OK;
    }
    int kv_join() {
        if (u.native_comm) return uv_comm_kv_wait(u.native_comm, s);
        int rc = u.kv_exchange(u.comm_user, kvs.o_send, kvs.o_first, kvs.o_prev, kvs.o_rfirst, kvs.nbytes);
        if (rc) {
            uv_set_error("kv_exchange callback failed (%d)", rc);
            return UV_ERR_STATE;
        }
        if (kvs.emu && u.rank > 0) RUN(uv_comm_launch_wait(u.emu_flag, u.emu_epoch, (int*)(u.emu_flag + 1), s));
        return UV_OK;
    }

    // resnet.py:335-394
    int resblock(const std::string& p, const Act& x, const Act* skip, int Cout, Act* out) {
        const int Cin = x.C + (skip ? skip->C : 0);
        const int rps = F * x.H * x.W;
        half_t* n1 = alloc(x.rows() * Cin);
        if (!n1) return UV_ERR_STATE;
        RUN(groupnorm(x, skip, rps, u.cfg.norm_eps, p + ".norm1", 1, n1, true));
        auto to = u.temb_off.find(p);
        UV_REQUIRE(to != u.temb_off.end() && temb_all, "%s: time_emb_proj missing", p.c_str());
        Act n1a{n1, x.imgs, x.H, x.W, Cin}, h;
        RUN(conv(n1a, nullptr, p + ".conv1", Cout, 9, 1, 0, temb_all + to->second, nullptr, &h, u.temb_total, true));
        free(n1);
        half_t* n2 = alloc(h.rows() * Cout);
        if (!n2) return UV_ERR_STATE;
        RUN(groupnorm(h, nullptr, rps, u.cfg.norm_eps, p + ".norm2", 1, n2, true));
        free(h.p);
        const half_t* res = x.p;
        Act sc{};
        if (u.find(p + ".conv_shortcut.weight")) {
            RUN(conv(x, skip, p + ".conv_shortcut", Cout, 1, 1, 0, nullptr, nullptr, &sc));
            res = sc.p;
        } else {
            UV_REQUIRE(!skip && x.C == Cout, "%s: no conv_shortcut but channel mismatch", p.c_str());
        }
        Act n2a{n2, x.imgs, x.H, x.W, Cout};
        RUN(conv(n2a, nullptr, p + ".conv2", Cout, 9, 1, 0, nullptr, res, out, 0, true));
        free(n2);
        if (sc.p) free(sc.p);
        return UV_OK;
    }

    // attention.py:104-153 + :280-346 (+ pnp_utils.py:20-100 when pnp_layer)
    int transformer(const std::string& p, const Act& x, bool pnp_layer, int heads, Act* out) {
        const int C = x.C, d = C / heads, N = x.H * x.W;
        const long rows = x.rows();
        const std::string b = p + ".transformer_blocks.0";
        half_t* t0 = alloc(rows * C);
        if (!t0) return UV_ERR_STATE;
        // The block's per-frame GroupNorm (attention.py:121) folded into proj_in where that pays (round 5): per frame a weight set gamma * rstd (.) W and
        // an fp32 bias carrying the mean term, proj_in then reads the RAW tensor — the apply pass (read + write of the tensor) goes for B*F weight copies
        // (9.8 MB against 252 MB at the 64x64 level; at the 32x32 level the copies are a third of the pass: not taken)
        const std::string wpi = p + (u.find(p + ".proj_in.weight#nhwc") ? ".proj_in.weight#nhwc" : ".proj_in.weight");
        const long set_bytes = (long)x.imgs * C * C * 2, apply_bytes = rows * C * 4;
        const bool gnf = u.gn_fold && uv_linear_takes_big_direct(rows, C, C) && (N % 256 == 0 || N % 192 == 0) && 8 * set_bytes <= apply_bytes;
        half_t* wsets = nullptr;
        float* b32 = nullptr;
        if (gnf) {
            wsets = alloc((long)x.imgs * C * C);
            b32 = (float*)alloc((long)x.imgs * C * 2);
            if (!wsets || !b32) return UV_ERR_STATE;
            UvGnFold gf;
            gf.W = W(wpi); gf.bias = W(p + ".proj_in.bias"); gf.N = C; gf.W_out = wsets; gf.bias32 = b32;
            if (!gf.W || !gf.bias) return u.missing_error();
            RUN(groupnorm(x, nullptr, N, 1e-6f, p + ".norm", 0, nullptr, false, &gf));
        } else {
            RUN(groupnorm(x, nullptr, N, 1e-6f, p + ".norm", 0, t0));
        }
        half_t* h = alloc(rows * C);
        if (!h) return UV_ERR_STATE;
        // The three LayerNorms of the block are folded into the linears around them when all of those take the direct 256x320 path
        // (levels with >= 150 tiles: the 64x64 .. 16x16 levels of an unsharded clip): the producing linear's epilogue leaves the row
        // statistics, the consuming one runs on the raw rows — no LayerNorm launch, no normalised copy in HBM.
        // norm3 -> GEGLU projection is folded only on request (ln_fold = 2): its epilogue is the longest of the block and the fold
        // costs it more than the LayerNorm launch it removes (tools/bench_ln_fold.py).
        // (round 4: also on the 128-wide kernel where that runs these linears without split-K — the 64x64 / 32x32 levels of a frame shard)
        const bool fold = u.ln_fold > 0 && C % 160 == 0 && uv_linear_fold_producer_ok(rows, C, C) && uv_linear_fold_consumer_ok(rows, 3 * C, C, false) &&
                          uv_linear_fold_consumer_ok(rows, C, C, false);
        // norm3 with the statistics from a 128-wide producer: UNIVST_LN_FOLD_SMALL=1 leaves it a LayerNorm launch (A/B: emulated rank of 8,
        // F = 16: 12.83 ms per step without the 128-wide fold, 12.81 with norm1 / norm2 only, 12.70 with all three)
        static const int fold_small = getenv("UNIVST_LN_FOLD_SMALL") ? atoi(getenv("UNIVST_LN_FOLD_SMALL")) : 2;
        const bool xres3 = u.find(b + ".ff.net.0.proj.weight#xres") != nullptr && uv_geglu_xres_ok(8 * C, C, rows);
        const bool fold3 = fold && u.ln_fold > 1 && (xres3 || uv_linear_fold_consumer_ok(rows, 8 * C, C, true)) && (fold_small > 1 || uv_linear_takes_big_direct(rows, C, C));
        float* lnst = fold ? (float*)alloc(rows * (C / 160) * 4) : nullptr;      // [rows][C/160][2] fp32
        if (fold && !lnst) return UV_ERR_STATE;
        RUN(linear(gnf ? x.p : t0, C, rows, C, wpi, p + ".proj_in.bias", C, h, C, nullptr, 0, nullptr, 0, lnst, nullptr, nullptr, wsets, b32, N));
        if (gnf) {
            free(wsets);
            free(b32);
        }
        // ---- attn1
        half_t *gm, *bt;
        gm = W(b + ".norm1.weight"); bt = W(b + ".norm1.bias");
        if (!gm || !bt) return u.missing_error();
        if (!fold) RUN(uv_launch_layernorm(h, C, t0, C, gm, bt, rows, C, 1e-5f, s));
        const bool shard = u.world > 1, halo = shard && u.rank > 0;
        if (shard) RUN(kv_post(h, C, N));                  // the boundary frames' hidden rows leave now, on the forked stream
        const long extra_rows = shard ? (long)2 * B * N : 0;     // the halo frames' q|k|v rows behind the local ones: [prev: B x N | first: B x N]
        half_t* qkv = alloc((rows + extra_rows) * 3 * C);
        if (!qkv) return UV_ERR_STATE;
        const bool qs = u.find(b + ".attn1.qkv#fused#qs") != nullptr;      // head_dim 40: scale folded into to_q (finalize)
        const std::string wqkv = b + (qs ? ".attn1.qkv#fused#qs" : ".attn1.qkv#fused");
        if (fold) RUN(linear(h, C, rows, C, wqkv + "#ln", "", 3 * C, qkv, 3 * C, nullptr, 0, nullptr, 0, nullptr, lnst));
        else RUN(linear(t0, C, rows, C, wqkv, "", 3 * C, qkv, 3 * C));
        const bool registered = pnp_layer && pnp && pnp->registered;
        const bool shift = registered && pnp->idx >= pnp->eta1 && pnp->idx <= pnp->eta2 * 50.f;
        float beta = 0.f;
        if (shift) {
            UV_REQUIRE(B == 3, "PnP attention shift needs the three-branch batch (B=3), got B=%d", B);
            beta = (0.9f - 0.1f) / (pnp->eta1 * 50.f - pnp->eta2 * 50.f) * ((float)pnp->idx - pnp->eta2 * 50.f) + 0.1f;
            RUN(uv_launch_adain_shift(qkv, 3 * C, F, N, C, ad_ws, ad_ws + (long)F * 2 * C, pnp->alpha, beta, pnp->gamma, s));
        }
        AttnParams ap;
        ap.q = qkv; ap.k = qkv + C; ap.v = qkv + 2 * C;
        ap.ldq = ap.ldkv = 3 * C;
        ap.o = t0; ap.ldo = C;
        ap.src_idx = registered ? idx_pnp : idx_stock;
        ap.src_cnt = registered ? cnt_pnp : cnt_stock;
        ap.src_logw = registered ? lw_pnp : lw_stock;
        ap.nsrc = registered ? 2 : 3;
        ap.BF = x.imgs; ap.Nq = N; ap.Nkv = N; ap.heads = heads; ap.d = d;
        ap.scale_log2e = 1.4426950408889634f / sqrtf((float)d);
        ap.q_prescaled = qs;
        float* state = nullptr;
        if (halo) {               // phase 1: the key frames this rank holds; (m, l) per query stays behind for phase 2
            state = (float*)alloc((long)x.imgs * heads * N * 4);
            if (!state) return UV_ERR_STATE;
            ap.state_out = state;
        }
        RUN(uv_launch_attention(ap, s));
        if (shard) RUN(kv_join());
        if (halo) {
            // the two halo frames: LayerNorm (norm1) of the received hidden rows -> to_k | to_v (all of q|k|v inside the window: the shift kernel works on
            // the fused rows) -> the AdaIN shift of each frame -> phase 2 over them.  The sender would have computed the same K | V from the same fp16 rows
            // (with the LayerNorm folded into the GEMM when `fold`: equal up to fp16 rounding of the normalised rows).
            const long hrows = 2L * B * N;
            half_t* tn = alloc(hrows * C);
            if (!tn) return UV_ERR_STATE;
            RUN(uv_launch_layernorm((const half_t*)(u.comm_ws + kvs.o_prev), C, tn, C, gm, bt, (long)B * N, C, 1e-5f, s));
            RUN(uv_launch_layernorm((const half_t*)(u.comm_ws + kvs.o_rfirst), C, tn + (long)B * N * C, C, gm, bt, (long)B * N, C, 1e-5f, s));
            half_t* qh = qkv + rows * 3 * C;
            half_t* wfull = W(wqkv);
            if (!wfull) return u.missing_error();
            if (shift) {
                RUN(linear(tn, C, hrows, C, wqkv, "", 3 * C, qh, 3 * C));
                for (int sl = 0; sl < 2; ++sl)            // rows [slot][3 branches][N] = the shift kernel's [3][F = 1][N]
                    RUN(uv_launch_adain_shift(qh + (long)sl * B * N * 3 * C, 3 * C, 1, N, C, ad_ws, ad_ws + 2L * C, pnp->alpha, beta, pnp->gamma, s));
            } else {
                RUN(linear(tn, C, hrows, C, wqkv, "", 2 * C, qh + C, 3 * C, nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, wfull + (long)C * C));
            }
            free(tn);
            ap.src_idx = registered ? idx2_pnp : idx2_stock;
            ap.src_cnt = registered ? cnt2_pnp : cnt2_stock;
            ap.src_logw = nullptr;
            ap.state_out = nullptr;
            ap.state_in = state;
            RUN(uv_launch_attention(ap, s));
            ap.state_in = nullptr;
            free(state);
        }
        free(qkv);
        // ---- the row-local chain behind the self-attention: to_out + residual -> (norm2) -> attn2 -> to_out + residual -> (norm3) ->
        // GEGLU projection -> FF2 + residual.  Round 4: it runs BAND BY BAND over whole frames when the level is large (64x64 level of
        // an unsharded clip: 196 608 rows, every C-wide tensor 126 MB, the 4C-wide hidden one 503 MB): with bands of 65 536 rows (one
        // round of 256-row tiles on the 256 CUs) the band's intermediates (42 MB each, hidden 168 MB) are re-read by the next kernel of
        // the chain while they are still in the 256 MB Infinity Cache instead of streaming from HBM, and the hidden buffer is
        // band-sized (tools/bench_mall_bands.py: -13 % on the four FF linears; every kernel is row-local: same results up to fp32 summation order).
        // OFF by default: inside the step (same box, alternating runs) the banded graph is 0.4 ms SLOWER (linears 16.7 -> 17.0 ms: three
        // one-round launches per kernel lose more to launch tails than the cache returns).  Kept as a switch (chain_bands).
        const long N_rows = N;
        int nbands = 1;
        if (u.chain_bands != 1) {
            const long target = 65536;
            int want = u.chain_bands > 1 ? u.chain_bands : (int)(rows / target);
            while (want > 1 && (x.imgs % want != 0 || (u.chain_bands <= 1 && rows / want < target))) --want;
            nbands = want < 1 ? 1 : want;
            // the folded LayerNorms need the direct 256x320 path for the band's row count as well: a level too small for that stays unbanded
            if (nbands > 1 && fold && !(uv_linear_takes_big_direct(rows / nbands, C, C) && (!fold3 || uv_linear_takes_big_direct(rows / nbands, 8 * C, C)))) nbands = 1;
        }
        const int band_imgs = x.imgs / nbands;
        const long brows = (long)band_imgs * N_rows;
        // (buffers are taken when first needed and, with ONE band, released as soon as the chain is past them — the arena's
        // high-water mark of the unbanded graph is unchanged; with bands the full-height tensors live until the last band, the band-sized
        // q2 / hidden buffers are reused by every band)
        half_t* h2 = alloc(rows * C);
        half_t* kv = alloc((long)B * text_len * 2 * C);
        half_t *h3 = nullptr, *h4 = nullptr, *q2 = nullptr, *mid = nullptr;
        if (!h2 || !kv) return UV_ERR_STATE;
        half_t *gm2 = W(b + ".norm2.weight"), *bt2 = W(b + ".norm2.bias"), *gm3 = W(b + ".norm3.weight"), *bt3 = W(b + ".norm3.bias");
        half_t* tb = W(b + ".attn_temporal.to_out.0.bias");
        if (!gm2 || !bt2 || !gm3 || !bt3 || !tb) return u.missing_error();
        const bool qs2 = u.find(b + ".attn2.to_q.weight#qs") != nullptr;
        const std::string wq2 = b + (qs2 ? ".attn2.to_q.weight#qs" : ".attn2.to_q.weight");
        const bool t_attn = u.temporal_attn_active.count(b) != 0;
        RUN(linear(text, u.cfg.cross_attention_dim, (long)B * text_len, u.cfg.cross_attention_dim, b + ".attn2.kv#fused", "",
                   2 * C, kv, 2 * C));
        // round 5, second step: the self-attention's out projection rides in FRONT of the fused text cross-attention (fused.hip, PRE): h2 = to_out(attn1) + h
        // is that kernel's input and residual and nothing else reads it, so it stays in the block's LDS
        const std::string wqf0 = wq2 + "#ln#frag", wof0 = b + ".attn2.to_out.0.weight#frag", wpf0 = b + ".attn1.to_out.0.weight#frag";
        const bool a2pre = nbands == 1 && fold && u.attn2_fused > 1 && uv_attn2_fused_ok(C, heads, F * N, text_len) && u.find(wqf0) && u.find(wof0) && u.find(wpf0);
        for (int bd = 0; bd < nbands; ++bd) {
            const long r0 = (long)bd * brows, o = r0 * C;
            float* lb = lnst ? lnst + r0 * (C / 160) * 2 : nullptr;
            if (a2pre) {
                half_t* kvf = alloc(uv_attn2_kvf_halfs(B, heads, d));
                if (!kvf || !(h3 = alloc(rows * C))) return UV_ERR_STATE;
                RUN(uv_launch_kv_frag_pack(kv, kvf, B, text_len, C, heads, s));
                Attn2Params a2;
                a2.X = t0; a2.ldx = C; a2.M = (int)rows;
                a2.Wp_f = W(wpf0); a2.bias_p = W(b + ".attn1.to_out.0.bias"); a2.Rp = h; a2.ldrp = C;
                a2.ln_eps = 1e-5f; a2.ln_slots = C / 160;
                a2.ln_wsum = (const float*)W(wq2 + "#ln.wsum"); a2.ln_bias = (const float*)W(wq2 + "#ln.bias");
                a2.Wq_f = W(wqf0); a2.kvf = kvf;
                a2.rows_per_branch = F * N; a2.heads = heads; a2.Nkv = text_len;
                a2.q_prescaled = qs2; a2.scale_log2e = ap.scale_log2e;
                a2.Wo_f = W(wof0); a2.bias_o = W(b + ".attn2.to_out.0.bias");
                a2.Y = h3; a2.ldy = C;
                a2.stats_out = fold3 ? lb : nullptr;
                if (!a2.bias_p || !a2.bias_o || !a2.ln_wsum || !a2.ln_bias) return u.missing_error();
                RUN(uv_launch_attn2_fused(a2, C, s));
                free(kvf);
                free(kv);
                free(h);
                free(h2);
            } else {
            RUN(linear(t0 + o, C, brows, C, b + ".attn1.to_out.0.weight", b + ".attn1.to_out.0.bias", C, h2 + o, C, h + o, C, nullptr, 0, lb));
            if (nbands == 1) free(h);
            // ---- attn2 (text)
            // ONE launch where the shape allows (round 5, fused.hip): q projection (LayerNorm folded), the 77-key attention of the wave's two
            // heads and the out projection + residual share one 64-row tile in LDS — Q and O never reach HBM
            const std::string wqf = wq2 + (fold ? "#ln#frag" : "#frag"), wof = b + ".attn2.to_out.0.weight#frag";
            if (nbands == 1 && u.attn2_fused && uv_attn2_fused_ok(C, heads, F * N, text_len) && u.find(wqf) && u.find(wof)) {
                half_t* kvf = alloc(uv_attn2_kvf_halfs(B, heads, d));
                if (!kvf || !(h3 = alloc(rows * C))) return UV_ERR_STATE;
                RUN(uv_launch_kv_frag_pack(kv, kvf, B, text_len, C, heads, s));
                if (!fold) RUN(uv_launch_layernorm(h2, C, t0, C, gm2, bt2, rows, C, 1e-5f, s));
                Attn2Params a2;
                a2.X = fold ? h2 : t0; a2.ldx = C; a2.M = (int)rows;
                if (fold) {
                    a2.ln_stats = lb; a2.ln_slots = C / 160; a2.ln_eps = 1e-5f;
                    a2.ln_wsum = (const float*)W(wq2 + "#ln.wsum"); a2.ln_bias = (const float*)W(wq2 + "#ln.bias");
                    if (!a2.ln_wsum || !a2.ln_bias) return u.missing_error();
                }
                a2.Wq_f = W(wqf); a2.kvf = kvf;
                a2.rows_per_branch = F * N; a2.heads = heads; a2.Nkv = text_len;
                a2.q_prescaled = qs2; a2.scale_log2e = ap.scale_log2e;
                a2.Wo_f = W(wof); a2.bias_o = W(b + ".attn2.to_out.0.bias");
                a2.R = h2; a2.ldr = C; a2.Y = h3; a2.ldy = C;
                a2.stats_out = fold3 ? lb : nullptr;
                if (!a2.bias_o) return u.missing_error();
                RUN(uv_launch_attn2_fused(a2, C, s));
                free(kvf);
                free(kv);
                free(h2);
            } else {
            if (!q2 && !(q2 = alloc(brows * C))) return UV_ERR_STATE;
            if (!fold) RUN(uv_launch_layernorm(h2 + o, C, t0 + o, C, gm2, bt2, brows, C, 1e-5f, s));
            if (fold) RUN(linear(h2 + o, C, brows, C, wq2 + "#ln", "", C, q2, C, nullptr, 0, nullptr, 0, nullptr, lb));
            else RUN(linear(t0 + o, C, brows, C, wq2, "", C, q2, C));
            ap.q = q2; ap.ldq = C;
            ap.k = kv; ap.v = kv + C; ap.ldkv = 2 * C;
            ap.o = t0 + o; ap.ldo = C;
            ap.src_idx = idx_text + bd * band_imgs; ap.src_cnt = nullptr; ap.src_logw = nullptr; ap.nsrc = 1; ap.Nkv = text_len;
            ap.BF = band_imgs;
            ap.q_prescaled = qs2;
            RUN(uv_launch_attention(ap, s));
            if (nbands == 1) { free(q2); free(kv); }
            if (!h3 && !(h3 = alloc(rows * C))) return UV_ERR_STATE;
            RUN(linear(t0 + o, C, brows, C, b + ".attn2.to_out.0.weight", b + ".attn2.to_out.0.bias", C, h3 + o, C, h2 + o, C, nullptr, 0, fold3 ? lb : nullptr));
            if (nbands == 1) free(h2);
            }
            }
            if (!mid && !(mid = alloc(brows * 4 * C))) return UV_ERR_STATE;
            // ---- GEGLU feed-forward (+ the bias-only temporal attention, attention.py:233)
            // (K = 320: the X-resident kernel and its weight order, any row count)
            const bool xres = u.find(b + ".ff.net.0.proj.weight#xres") != nullptr && uv_geglu_xres_ok(8 * C, C, brows);
            const std::string wff = b + (xres ? ".ff.net.0.proj.weight#xres" : ".ff.net.0.proj.weight#geglu");
            const std::string bff = b + (xres ? ".ff.net.0.proj.bias#xres" : ".ff.net.0.proj.bias#geglu");
            if (!fold3) {
                RUN(uv_launch_layernorm(h3 + o, C, t0 + o, C, gm3, bt3, brows, C, 1e-5f, s));
                RUN(linear(t0 + o, C, brows, C, wff, bff, 8 * C, mid, 4 * C, nullptr, 0, nullptr, xres ? 2 : 1));
            } else {
                RUN(linear(h3 + o, C, brows, C, wff + "#ln", "", 8 * C, mid, 4 * C, nullptr, 0, nullptr, xres ? 2 : 1, nullptr, lb));
            }
            if (nbands == 1 && lnst) free(lnst);
            if (!h4 && !(h4 = alloc(rows * C))) return UV_ERR_STATE;
            RUN(linear(mid, 4 * C, brows, 4 * C, b + ".ff.net.2.weight", b + ".ff.net.2.bias", C, h4 + o, C, h3 + o, C, t_attn ? nullptr : tb));
        }
        ap.BF = x.imgs;                // (the band loop narrowed it)
        if (nbands > 1) {
            free(h);
            free(q2);
            free(kv);
            free(h2);
            if (lnst) free(lnst);
        }
        free(mid);
        free(h3);
        if (t_attn) {      // TRAINED temporal attention (attention.py:336-346): LayerNorm, q|k|v, softmax over the F frames of every pixel, to_out + residual
            UV_REQUIRE(F <= 32, "temporal attention: clips of %d frames (kernel holds <= 32 scores per query)", F);
            gm = W(b + ".norm_temporal.weight"); bt = W(b + ".norm_temporal.bias");
            if (!gm || !bt) return u.missing_error();
            RUN(uv_launch_layernorm(h4, C, t0, C, gm, bt, rows, C, 1e-5f, s));
            half_t* qkvt = alloc(rows * 3 * C);
            if (!qkvt) return UV_ERR_STATE;
            RUN(linear(t0, C, rows, C, b + ".attn_temporal.qkv#fused", "", 3 * C, qkvt, 3 * C));
            const long nthr = (long)B * N * heads * F;
            const float sc = 1.f / sqrtf((float)d);
            switch (d) {
                case 8: hipLaunchKernelGGL((temporal_attn_kernel<1>), dim3(nb(nthr)), dim3(256), 0, s, qkvt, t0, B, F, N, C, heads, sc); break;
                case 16: hipLaunchKernelGGL((temporal_attn_kernel<2>), dim3(nb(nthr)), dim3(256), 0, s, qkvt, t0, B, F, N, C, heads, sc); break;
                case 32: hipLaunchKernelGGL((temporal_attn_kernel<4>), dim3(nb(nthr)), dim3(256), 0, s, qkvt, t0, B, F, N, C, heads, sc); break;
                case 40: hipLaunchKernelGGL((temporal_attn_kernel<5>), dim3(nb(nthr)), dim3(256), 0, s, qkvt, t0, B, F, N, C, heads, sc); break;
                case 64: hipLaunchKernelGGL((temporal_attn_kernel<8>), dim3(nb(nthr)), dim3(256), 0, s, qkvt, t0, B, F, N, C, heads, sc); break;
                case 80: hipLaunchKernelGGL((temporal_attn_kernel<10>), dim3(nb(nthr)), dim3(256), 0, s, qkvt, t0, B, F, N, C, heads, sc); break;
                case 160: hipLaunchKernelGGL((temporal_attn_kernel<20>), dim3(nb(nthr)), dim3(256), 0, s, qkvt, t0, B, F, N, C, heads, sc); break;
                default: uv_set_error("temporal attention: head_dim=%d not instantiated", d); return UV_ERR_UNSUPPORTED;
            }
            UV_LAUNCH_CHECK();
            free(qkvt);
            half_t* h5 = alloc(rows * C);
            if (!h5) return UV_ERR_STATE;
            RUN(linear(t0, C, rows, C, b + ".attn_temporal.to_out.0.weight", b + ".attn_temporal.to_out.0.bias", C, h5, C, h4, C));
            free(h4);
            h4 = h5;
        }
        free(t0);
        out->imgs = x.imgs; out->H = x.H; out->W = x.W; out->C = C;
        out->p = alloc(rows * C);
        if (!out->p) return UV_ERR_STATE;
        RUN(linear(h4, C, rows, C, p + (u.find(p + ".proj_out.weight#nhwc") ? ".proj_out.weight#nhwc" : ".proj_out.weight"), p + ".proj_out.bias", C, out->p,
                   C, x.p, C, nullptr, 0, nullptr, nullptr, &out->gst));
        free(h4);
        return UV_OK;
    }
};

int UNet::missing_error() {
    uv_set_error("weight '%s' was never loaded", missing.c_str());
    return UV_ERR_STATE;
}

int UNet::forward(const half_t* sample, float timestep, const half_t* text, int B, int F, int H, int Wd, int text_len,
                  const univst_pnp_t* pnp, half_t* eps_out, half_t* feat_out, int ft_index, hipStream_t s) {
    UV_REQUIRE(finalized, "forward: call univst_unet_finalize after loading weights");
    UV_REQUIRE(world == 1 || (temporal_conv_active.empty() && temporal_attn_active.empty()),
               "forward: trained temporal layers couple all frames of a pixel; frame sharding (world=%d) is not supported with them", world);
    if (native_comm) {
        RUN(uv_comm_poll(native_comm));          // a peer timed out in an earlier call: report instead of queueing more work
        uv_comm_bind_stream(native_comm, s);
    }
    UV_REQUIRE(B >= 1 && B <= 8 && F >= 1 && H >= 8 && Wd >= 8 && H % 8 == 0 && Wd % 8 == 0,
               "forward: unsupported geometry B=%d F=%d H=%d W=%d (H, W multiples of 8; B <= 8)", B, F, H, Wd);
    RUN(reserve(B, F, H, Wd));
    missing.clear();
    const int* boc = cfg.block_out_channels;
    const int C0 = boc[0], TED = 4 * C0, L = cfg.layers_per_block;
    Fwd f{*this, s, B, F, text_len, pnp};
    f.text = text;
    const int* tab = idx_tables[((long)B << 20) | F | ((long)rank << 40) | ((long)world << 50)];
    const long BF_ = (long)B * F;
    f.idx_stock = tab;
    f.idx_pnp = tab + BF_ * 3;
    f.idx_text = tab + BF_ * 5;
    f.cnt_stock = tab + BF_ * 6;
    f.cnt_pnp = tab + BF_ * 7;
    f.lw_stock = (const float*)(tab + BF_ * 8);
    f.lw_pnp = (const float*)(tab + BF_ * 11);
    if (world > 1 && rank > 0) {
        f.idx2_stock = tab + BF_ * 13;
        f.idx2_pnp = tab + BF_ * 16;
        f.cnt2_stock = tab + BF_ * 18;
        f.cnt2_pnp = tab + BF_ * 19;
    }
    f.gn_ws = (float*)arena.alloc((size_t)uv_groupnorm_workspace_floats(B * F, cfg.norm_num_groups) * sizeof(float));
    f.ad_ws = (float*)arena.alloc((size_t)F * 2 * boc[3] * 2 * sizeof(float) + 1024);
    f.sk_ws = (float*)arena.alloc(UV_SPLITK_WS_BYTES);
    UV_REQUIRE(f.gn_ws && f.ad_ws && f.sk_ws, "forward: arena too small");
    if (gn_producer) {             // statistics of up to 64 level-0-sized tensors: [rows/16][G][2] fp32 each (3 MB per tensor at 3 x 16 x 64 x 64: 201 MB, counted in
                                   // reserve()); taken AFTER the mandatory workspaces, so it is the part that degrades first (consumers fall back to their own pass)
        f.gst_left = (size_t)(((long)B * F * H * Wd + 15) / 16) * cfg.norm_num_groups * 2 * 64;
        f.gst_pool = (float*)arena.alloc(f.gst_left * sizeof(float));
        if (!f.gst_pool) f.gst_left = 0;
    }

    // ---- time embedding (unet_3d_condition.py:359-365)
    half_t* tsin = f.alloc((long)B * C0);
    half_t* e1 = f.alloc((long)B * TED);
    f.emb = f.alloc((long)B * TED);
    if (!tsin || !e1 || !f.emb) return UV_ERR_STATE;
    RUN(uv_launch_timestep_embed(timestep, tsin, B, C0, cfg.flip_sin_to_cos, cfg.freq_shift, s));
    {
        half_t *w1 = W("time_embedding.linear_1.weight"), *b1 = W("time_embedding.linear_1.bias");
        half_t *w2 = W("time_embedding.linear_2.weight"), *b2 = W("time_embedding.linear_2.bias");
        if (!w1 || !b1 || !w2 || !b2) return missing_error();
        RUN(uv_launch_linear_small(tsin, w1, b1, e1, B, TED, C0, 0, s));
        RUN(uv_launch_linear_small(e1, w2, b2, f.emb, B, TED, TED, 1, s));
        if (temb_total > 0) {           // resnet.py:349-351 for all resnets at once
            half_t *wa = W("time_emb_proj#all.weight"), *ba = W("time_emb_proj#all.bias");
            f.temb_all = f.alloc((long)B * temb_total);
            if (!wa || !ba || !f.temb_all) return missing_error();
            RUN(uv_launch_linear_small(f.emb, wa, ba, f.temb_all, B, (int)temb_total, TED, 1, s));
        }
    }
    // ---- conv_in
    const int CP = (cfg.in_channels + 7) / 8 * 8;
    Act x0{f.alloc((long)B * F * H * Wd * CP), B * F, H, Wd, CP};
    if (!x0.p) return UV_ERR_STATE;
    RUN(uv_launch_ncfhw_to_nhwc(sample, x0.p, B, cfg.in_channels, F, H * Wd, CP, s));
    Act x;
    RUN(f.conv(x0, nullptr, "conv_in", C0, 9, 1, 0, nullptr, nullptr, &x, 0, true));
    f.free(x0.p);

    std::vector<Act> skips;
    skips.push_back(x);
    // ---- down
    for (int i = 0; i < 4; ++i) {
        const std::string p = "down_blocks." + std::to_string(i);
        const bool has_attn = i < 3;
        for (int j = 0; j < L; ++j) {
            Act y;
            RUN(f.resblock(p + ".resnets." + std::to_string(j), x, nullptr, boc[i], &y));
            bool x_is_skip = false;
            for (auto& sk : skips) x_is_skip |= (sk.p == x.p);
            if (!x_is_skip) f.free(x.p);
            x = y;
            if (has_attn) {
                Act z;
                RUN(f.transformer(p + ".attentions." + std::to_string(j), x, false, cfg.attention_heads[i], &z));
                f.free(x.p);
                x = z;
            }
            skips.push_back(x);
        }
        if (i != 3) {
            Act y;
            RUN(f.conv(x, nullptr, p + ".downsamplers.0.conv", boc[i], 9, 2, 0, nullptr, nullptr, &y, 0, true));
            x = y;
            skips.push_back(x);
        }
    }
    // ---- mid
    {
        Act y, z, w;
        RUN(f.resblock("mid_block.resnets.0", x, nullptr, boc[3], &y));
        RUN(f.transformer("mid_block.attentions.0", y, false, cfg.attention_heads[3], &z));
        f.free(y.p);
        RUN(f.resblock("mid_block.resnets.1", z, nullptr, boc[3], &w));
        f.free(z.p);
        x = w;   // previous x is skips.back(), still owned by the skip list
    }
    // ---- up
    static const int pnp_layers[4][3] = {{0, 0, 0}, {0, 1, 1}, {1, 1, 1}, {1, 1, 1}};   // pnp_utils.py:104-111
    for (int i = 0; i < 4; ++i) {
        const std::string p = "up_blocks." + std::to_string(i);
        const int Cout = boc[3 - i];
        const bool has_attn = i > 0;
        for (int j = 0; j < L + 1; ++j) {
            Act sk = skips.back();
            skips.pop_back();
            Act y;
            RUN(f.resblock(p + ".resnets." + std::to_string(j), x, &sk, Cout, &y));
            f.free(x.p);
            f.free(sk.p);
            x = y;
            if (has_attn) {
                Act z;
                RUN(f.transformer(p + ".attentions." + std::to_string(j), x, j < 3 && pnp_layers[i][j], cfg.attention_heads[3 - i], &z));
                f.free(x.p);
                x = z;
            }
        }
        if (i != 3) {
            Act y;
            RUN(f.conv(x, nullptr, p + ".upsamplers.0.conv", Cout, 9, 1, 1, nullptr, nullptr, &y, 0, true));
            f.free(x.p);
            x = y;
        }
        if (feat_out && i == ft_index)   // sample[0].permute(1,2,3,0) == the first F*H*W NHWC rows
            UV_HIP(hipMemcpyAsync(feat_out, x.p, (size_t)F * x.H * x.W * x.C * sizeof(half_t), hipMemcpyDeviceToDevice, s));
    }
    // ---- out
    half_t* n = f.alloc(x.rows() * x.C);
    if (!n) return UV_ERR_STATE;
    RUN(f.groupnorm(x, nullptr, F * x.H * x.W, cfg.norm_eps, "conv_norm_out", 1, n, true));
    Act na{n, x.imgs, x.H, x.W, x.C}, y;
    RUN(f.conv(na, nullptr, "conv_out", cfg.out_channels, 9, 1, 0, nullptr, nullptr, &y));
    RUN(uv_launch_nhwc_to_ncfhw(y.p, cfg.out_channels, eps_out, B, cfg.out_channels, F, H * Wd, s));
    if (x_dirty) {                 // the forked stream's posts of this forward complete before anything the caller queues behind it (and a capture can end)
        UV_HIP(hipEventRecord(ev_join, xstream));
        UV_HIP(hipStreamWaitEvent(s, ev_join, 0));
        x_dirty = false;
    }
    if (!missing.empty()) return missing_error();
    return UV_OK;
}
