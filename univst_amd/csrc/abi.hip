// extern "C" surface of libunivst_hip.so (include/univst.h).  Thin argument checking + dispatch.
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <new>

#include "../../include/univst.h"
#include "common.h"
#include "kernels.h"
#include "unet.h"
#include "vae.h"

static thread_local char g_err[1024] = "";
void uv_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* uv_get_error() { return g_err; }

#define H(p) ((const half_t*)(p))
#define HM(p) ((half_t*)(p))
#define S(p) ((hipStream_t)(p))

struct univst_unet {
    UNet impl;
};

extern "C" {

const char* univst_last_error(void) { return g_err; }
int univst_abi_version(void) { return UNIVST_ABI_VERSION; }
int univst_sd3_shift_window(int idx, double eta1, double eta2, int* active, float* beta) {
    UV_REQUIRE(active && beta, "sd3_shift_window: null argument");
    // pnp_utils.py:183-186 in double, as the reference's Python evaluates it (in fp32, 0.3f * 50 = 15.000001 and step 15 falls out of the window)
    const bool on = eta1 * 50 <= (double)idx && (double)idx <= eta2 * 50;
    *active = on ? 1 : 0;
    *beta = on ? (float)((0.9 - 0.1) / (eta1 * 50 - eta2 * 50) * ((double)idx - eta2 * 50) + 0.1) : 0.f;
    return UV_OK;
}

int univst_unet_create(const univst_unet_cfg* cfg, univst_unet** out) {
    UV_REQUIRE(cfg && out, "unet_create: null argument");
    UV_REQUIRE(cfg->norm_num_groups > 0 && cfg->layers_per_block > 0, "unet_create: bad config");
    for (int i = 0; i < 4; ++i) {
        int c = cfg->block_out_channels[i];
        const int hd = cfg->attention_heads[i];
        UV_REQUIRE(hd > 0 && c % 8 == 0 && c % cfg->norm_num_groups == 0 && c % hd == 0 && (c / hd) % 8 == 0,
                   "unet_create: block_out_channels[%d]=%d must be divisible by 8, groups and heads (head_dim multiple of 8)", i, c);
    }
    UV_REQUIRE(cfg->cross_attention_dim % 8 == 0, "unet_create: cross_attention_dim must be a multiple of 8");
    univst_unet* h = new (std::nothrow) univst_unet();
    UV_REQUIRE(h, "unet_create: out of host memory");
    h->impl.cfg = *cfg;
    if (const char* e = getenv("UNIVST_LN_FOLD")) h->impl.ln_fold = atoi(e) < 0 ? 0 : (atoi(e) > 2 ? 2 : atoi(e));
    if (const char* e = getenv("UNIVST_GN_PRODUCER")) h->impl.gn_producer = atoi(e) != 0;
    if (const char* e = getenv("UNIVST_GN_FOLD")) h->impl.gn_fold = atoi(e) != 0;
    if (const char* e = getenv("UNIVST_ATTN2_PRE")) h->impl.attn2_fused = atoi(e) ? 2 : 1;
    if (const char* e = getenv("UNIVST_CHAIN_BANDS")) h->impl.chain_bands = atoi(e) < 0 ? 0 : atoi(e);
    if (const char* e = getenv("UNIVST_KV_OVERLAP")) h->impl.kv_overlap = atoi(e) != 0;
    *out = h;
    return UV_OK;
}
int univst_unet_set_option(univst_unet* h, const char* name, int value) {
    UV_REQUIRE(h && name, "unet_set_option: null argument");
    if (!strcmp(name, "ln_fold")) {
        UV_REQUIRE(value >= 0 && value <= 2, "unet_set_option: ln_fold is 0, 1 or 2");
        h->impl.ln_fold = value;
        return UV_OK;
    }
    if (!strcmp(name, "gn_fold")) {
        h->impl.gn_fold = value != 0;
        return UV_OK;
    }
    if (!strcmp(name, "attn2_fused")) {
        UV_REQUIRE(value >= 0 && value <= 2, "unet_set_option: attn2_fused is 0, 1 or 2");
        h->impl.attn2_fused = value;
        return UV_OK;
    }
    if (!strcmp(name, "gn_producer")) {
        h->impl.gn_producer = value != 0;
        return UV_OK;
    }
    if (!strcmp(name, "chain_bands")) {
        UV_REQUIRE(value >= 0 && value <= 64, "unet_set_option: chain_bands is 0 (auto), 1 (off) or a band count <= 64");
        h->impl.chain_bands = value;
        return UV_OK;
    }
    if (!strcmp(name, "kv_overlap")) {
        h->impl.kv_overlap = value != 0;
        return UV_OK;
    }
    if (!strcmp(name, "emu_wire_gbps")) {
        UV_REQUIRE(value >= 0 && value <= 10000, "unet_set_option: emu_wire_gbps is a per-link rate in GB/s (0: off)");
        h->impl.emu_wire_gbps = value;
        return UV_OK;
    }
    if (!strcmp(name, "emu_wire_lat_us")) {
        UV_REQUIRE(value >= 0 && value <= 100000, "unet_set_option: emu_wire_lat_us is a latency in microseconds");
        h->impl.emu_wire_lat_us = value;
        return UV_OK;
    }
    uv_set_error("unet_set_option: unknown option '%s'", name);
    return UV_ERR_ARG;
}
int univst_unet_query(univst_unet* h, const char* name, double* out) {
    UV_REQUIRE(h && name && out, "unet_query: null argument");
    if (!strcmp(name, "emu_wire_us")) {          // modelled wire time issued since the last query (reading resets it)
        *out = h->impl.emu_wire_us;
        h->impl.emu_wire_us = 0.0;
        return UV_OK;
    }
    if (!strcmp(name, "arena_high_water")) {
        *out = (double)h->impl.arena.high_water;
        return UV_OK;
    }
    uv_set_error("unet_query: unknown quantity '%s'", name);
    return UV_ERR_ARG;
}
int univst_unet_destroy(univst_unet* h) {
    delete h;
    return UV_OK;
}
int univst_unet_load_tensor(univst_unet* h, const char* key, const void* p, int dtype, const int64_t* shape, int ndim, void* s) {
    UV_REQUIRE(h, "null handle");
    return h->impl.load_tensor(key, p, dtype, shape, ndim, S(s));
}
int univst_unet_finalize(univst_unet* h, void* s) {
    UV_REQUIRE(h, "null handle");
    return h->impl.finalize(S(s));
}
int univst_unet_reserve(univst_unet* h, int B, int F, int Hh, int W) {
    UV_REQUIRE(h, "null handle");
    return h->impl.reserve(B, F, Hh, W);
}
int univst_unet_forward(univst_unet* h, const void* sample, float t, const void* text, int B, int F, int Hh, int W, int text_len,
                        const univst_pnp* pnp, void* eps, void* feat, int ft_index, void* s) {
    UV_REQUIRE(h && sample && text && eps, "unet_forward: null argument");
    return h->impl.forward(H(sample), t, H(text), B, F, Hh, W, text_len, pnp, HM(eps), HM(feat), ft_index, S(s));
}
int univst_unet_set_comm(univst_unet* h, int rank, int world, void* comm_ws, int64_t comm_ws_bytes, univst_allreduce_fn ar,
                         univst_kv_exchange_fn kv, void* user) {
    UV_REQUIRE(h && world >= 1 && rank >= 0 && rank < world, "set_comm: bad rank/world");
    UV_REQUIRE(world == 1 || (comm_ws && comm_ws_bytes >= (1 << 17) && ar && kv), "set_comm: world > 1 needs a workspace and both callbacks");
    h->impl.comm_ws = (char*)comm_ws;
    h->impl.comm_ws_bytes = comm_ws_bytes;
    h->impl.rank = rank;
    h->impl.world = world;
    h->impl.allreduce = ar;
    h->impl.kv_exchange = kv;
    h->impl.comm_user = user;
    h->impl.native_comm = nullptr;          // callbacks (or none, world == 1: detached) replace a library communicator
    return UV_OK;
}

int univst_unet_set_comm_native(univst_unet* h, univst_comm* comm) {
    UV_REQUIRE(h && comm, "set_comm_native: null argument");
    return uv_unet_attach_comm(h->impl, comm);
}

int univst_linear(const void* X, int64_t ldx, const void* W, const void* bias, const void* R, int64_t ldr, void* Y, int64_t ldy,
                  int M, int N, int K, int geglu, void* s) {
    UV_REQUIRE(X && W && Y, "linear: null argument");
    GemmParams g;
    g.X = H(X); g.ldx = ldx; g.W = H(W); g.bias = H(bias); g.R = H(R); g.ldr = ldr; g.Y = HM(Y); g.ldy = ldy;
    g.M = M; g.N = N; g.K = K; g.geglu = geglu;
    return uv_launch_gemm(g, 0, S(s));
}
int univst_linear_gated(const void* X, int64_t ldx, const void* W, const void* bias, const void* R, int64_t ldr, void* Y, int64_t ldy,
                        int M, int N, int K, int act, const void* gate, int64_t ld_gate, int rows_per_gate, void* s) {
    UV_REQUIRE(X && W && Y, "linear_gated: null argument");
    UV_REQUIRE(act == UNIVST_ACT_NONE || act == UNIVST_ACT_GELU_TANH, "linear_gated: act must be UNIVST_ACT_NONE or UNIVST_ACT_GELU_TANH");
    GemmParams g;
    g.X = H(X); g.ldx = ldx; g.W = H(W); g.bias = H(bias); g.R = H(R); g.ldr = ldr; g.Y = HM(Y); g.ldy = ldy;
    g.M = M; g.N = N; g.K = K;
    g.act = act == UNIVST_ACT_GELU_TANH ? 1 : 0;
    g.gate = H(gate); g.ld_gate = ld_gate; g.rows_per_gate = gate ? rows_per_gate : 1;
    return uv_launch_gemm(g, 0, S(s));
}
int univst_geglu_xres_permute(const void* in, void* out, int rows, int cols, void* s) {
    UV_REQUIRE(in && out, "geglu_xres_permute: null argument");
    return uv_launch_geglu_xres_permute(H(in), HM(out), rows, cols, S(s));
}
struct univst_vae {
    Vae impl;
};
int univst_vae_create(const univst_vae_cfg* cfg, univst_vae** out) {
    UV_REQUIRE(cfg && out, "vae_create: null argument");
    UV_REQUIRE(cfg->norm_num_groups > 0 && cfg->layers_per_block >= 1 && cfg->in_channels >= 1 && cfg->in_channels <= 8 && cfg->out_channels >= 1 &&
               cfg->out_channels <= 8 && cfg->latent_channels >= 1 && cfg->latent_channels <= 8, "vae_create: bad config");
    for (int i = 0; i < 4; ++i) {
        const int c = cfg->block_out_channels[i];
        UV_REQUIRE(c > 0 && c % 8 == 0 && c % cfg->norm_num_groups == 0 && (c / cfg->norm_num_groups) % 2 == 0,
                   "vae_create: block_out_channels[%d]=%d must be a multiple of 8 and an even multiple of the group count", i, c);
    }
    univst_vae* h = new (std::nothrow) univst_vae();
    UV_REQUIRE(h, "vae_create: out of host memory");
    h->impl.cfg = *cfg;
    *out = h;
    return UV_OK;
}
int univst_vae_destroy(univst_vae* h) {
    delete h;
    return UV_OK;
}
int univst_vae_load_tensor(univst_vae* h, const char* key, const void* p, int dtype, const int64_t* shape, int ndim, void* s) {
    UV_REQUIRE(h, "null handle");
    return h->impl.load_tensor(key, p, dtype, shape, ndim, S(s));
}
int univst_vae_finalize(univst_vae* h, void* s) {
    UV_REQUIRE(h, "null handle");
    return h->impl.finalize(S(s));
}
int univst_vae_decode(univst_vae* h, const void* z, int64_t imgs, int num_frames, int lat_h, int lat_w, void* out, void* s) {
    UV_REQUIRE(h && z && out, "vae_decode: null argument");
    return h->impl.decode(H(z), imgs, num_frames, lat_h, lat_w, HM(out), S(s));
}
int univst_vae_encode(univst_vae* h, const void* x, int64_t imgs, int Hh, int W, void* moments, void* s) {
    UV_REQUIRE(h && x && moments, "vae_encode: null argument");
    return h->impl.encode(H(x), imgs, Hh, W, HM(moments), S(s));
}
int univst_frag_pack(const void* W, void* out, int N, int K, void* s) {
    UV_REQUIRE(W && out, "frag_pack: null argument");
    return uv_launch_frag_pack(H(W), HM(out), N, K, S(s));
}
int64_t univst_attn2_fused_workspace_bytes(int B, int heads, int head_dim) { return uv_attn2_kvf_halfs(B, heads, head_dim) * (int64_t)sizeof(half_t); }
int univst_attn2_fused(const void* X, int64_t ldx, const float* ln_stats, float ln_eps, const float* ln_wsum, const float* ln_bias, const void* Wq_frag,
                       int q_prescaled, const void* kv, int B, int T, int64_t rows_per_branch, const void* Wo_frag, const void* bias_o,
                       const void* residual, int64_t ldr, void* Y, int64_t ldy, int64_t M, int C, int heads, float* stats_out, void* workspace,
                       void* s) {
    UV_REQUIRE(X && Wq_frag && kv && Wo_frag && residual && Y && workspace, "attn2_fused: null argument");
    UV_REQUIRE(B >= 1 && M >= 1 && M <= (int64_t)B * rows_per_branch && M < (1LL << 31) && C % 160 == 0 && heads > 0 && C % heads == 0,
               "attn2_fused: M=%lld rows exceed B=%d branches of %lld rows (or bad C / heads)", (long long)M, B, (long long)rows_per_branch);
    UV_REQUIRE(!ln_stats || (ln_wsum && ln_bias), "attn2_fused: a folded LayerNorm needs ln_wsum and ln_bias");
    UV_REQUIRE(uv_attn2_fused_ok(C, heads, (int)rows_per_branch, T), "attn2_fused: C=%d heads=%d rows_per_branch=%lld keys=%d is not a shape this kernel serves "
               "(C = 320, 8 heads, rows per branch a multiple of 64, <= 80 keys)", C, heads, (long long)rows_per_branch, T);
    int rc = uv_launch_kv_frag_pack(H(kv), HM(workspace), B, T, C, heads, S(s));
    if (rc) return rc;
    Attn2Params p;
    p.X = H(X); p.ldx = ldx; p.M = (int)M;
    p.ln_stats = ln_stats; p.ln_slots = C / 160; p.ln_eps = ln_eps; p.ln_wsum = ln_wsum; p.ln_bias = ln_bias;
    p.Wq_f = H(Wq_frag); p.kvf = H(workspace);
    p.rows_per_branch = (int)rows_per_branch; p.heads = heads; p.Nkv = T;
    p.q_prescaled = q_prescaled; p.scale_log2e = 1.4426950408889634f / sqrtf((float)(C / heads));
    p.Wo_f = H(Wo_frag); p.bias_o = H(bias_o);
    p.R = H(residual); p.ldr = ldr;
    p.Y = HM(Y); p.ldy = ldy;
    p.stats_out = stats_out;
    return uv_launch_attn2_fused(p, C, S(s));
}
int univst_attn12_fused(const void* attn_out, int64_t ldx, const void* Wp_frag, const void* bias_p, const void* residual_in, int64_t ldr_in, float ln_eps,
                        const float* ln_wsum, const float* ln_bias, const void* Wq_frag, int q_prescaled, const void* kv, int B, int T, int64_t rows_per_branch,
                        const void* Wo_frag, const void* bias_o, void* Y, int64_t ldy, int64_t M, int C, int heads, float* stats_out, void* workspace, void* s) {
    UV_REQUIRE(attn_out && Wp_frag && residual_in && ln_wsum && ln_bias && Wq_frag && kv && Wo_frag && Y && workspace, "attn12_fused: null argument");
    UV_REQUIRE(B >= 1 && M >= 1 && M <= (int64_t)B * rows_per_branch && M < (1LL << 31) && heads > 0 && C % heads == 0, "attn12_fused: M=%lld rows exceed B=%d branches of %lld rows",
               (long long)M, B, (long long)rows_per_branch);
    UV_REQUIRE(uv_attn2_fused_ok(C, heads, (int)rows_per_branch, T), "attn12_fused: C=%d heads=%d rows_per_branch=%lld keys=%d is not a shape this kernel serves "
               "(C = 320, 8 heads, rows per branch a multiple of 64, <= 80 keys)", C, heads, (long long)rows_per_branch, T);
    int rc = uv_launch_kv_frag_pack(H(kv), HM(workspace), B, T, C, heads, S(s));
    if (rc) return rc;
    Attn2Params p;
    p.X = H(attn_out); p.ldx = ldx; p.M = (int)M;
    p.Wp_f = H(Wp_frag); p.bias_p = H(bias_p); p.Rp = H(residual_in); p.ldrp = ldr_in;
    p.ln_slots = C / 160; p.ln_eps = ln_eps; p.ln_wsum = ln_wsum; p.ln_bias = ln_bias;
    p.Wq_f = H(Wq_frag); p.kvf = H(workspace);
    p.rows_per_branch = (int)rows_per_branch; p.heads = heads; p.Nkv = T;
    p.q_prescaled = q_prescaled; p.scale_log2e = 1.4426950408889634f / sqrtf((float)(C / heads));
    p.Wo_f = H(Wo_frag); p.bias_o = H(bias_o);
    p.Y = HM(Y); p.ldy = ldy;
    p.stats_out = stats_out;
    return uv_launch_attn2_fused(p, C, S(s));
}
int univst_linear_ln(const void* X, int64_t ldx, const void* W, const void* bias, const void* R, int64_t ldr, void* Y, int64_t ldy,
                     int M, int N, int K, int geglu, const float* ln_stats, float ln_eps, const float* ln_wsum, const float* ln_bias,
                     float* stats_out, void* s) {
    UV_REQUIRE(X && W && Y, "linear_ln: null argument");
    UV_REQUIRE(!ln_stats || (ln_wsum && ln_bias && K % 160 == 0 && !bias), "linear_ln: a folded LayerNorm needs wsum, lnb (which holds the bias) and K %% 160 == 0");
    UV_REQUIRE((!stats_out || uv_linear_fold_producer_ok(M, N, K)) && (!ln_stats || geglu == 2 || uv_linear_fold_consumer_ok(M, N, K, geglu != 0)),
               "linear_ln: M=%d N=%d K=%d is not taken by the direct 256x320 path (N %% 320 == 0 and >= 150 tiles) nor by the 128-wide path without split-K", M, N, K);
    GemmParams g;
    g.X = H(X); g.ldx = ldx; g.W = H(W); g.bias = H(bias); g.R = H(R); g.ldr = ldr; g.Y = HM(Y); g.ldy = ldy;
    g.M = M; g.N = N; g.K = K; g.geglu = geglu;
    g.ln_stats = ln_stats; g.ln_slots = K / 160; g.ln_eps = ln_eps; g.ln_wsum = ln_wsum; g.ln_bias = ln_bias; g.stats_out = stats_out;
    return uv_launch_gemm(g, 0, S(s));
}
int univst_conv_nhwc(const void* X1, const void* X2, int C1, int C2, int imgs, int Hs, int Ws, int up, int stride, int taps,
                     const void* W, const void* bias, const void* rowbias, int rows_per_rb, const void* R, void* Y, int Cout,
                     void* s) {
    UV_REQUIRE(X1 && W && Y, "conv: null argument");
    UV_REQUIRE((X2 != nullptr) == (C2 > 0), "conv: X2/C2 mismatch");
    UV_REQUIRE(stride == 1 || stride == 2, "conv: stride %d", stride);
    GemmParams g;
    g.X = H(X1); g.X2 = H(X2); g.C1 = C1; g.C2 = C2; g.Hs = Hs; g.Ws = Ws; g.up = up ? 1 : 0; g.stride = stride; g.taps = taps;
    const int He = Hs << g.up, We = Ws << g.up;
    g.Ho = taps == 9 ? (He - 1) / stride + 1 : He;
    g.Wo = taps == 9 ? (We - 1) / stride + 1 : We;
    g.M = imgs * g.Ho * g.Wo; g.N = Cout; g.K = taps * (C1 + C2);
    g.W = H(W); g.bias = H(bias); g.rowbias = H(rowbias); g.rows_per_rb = rows_per_rb > 0 ? rows_per_rb : 1;
    g.R = H(R); g.ldr = Cout; g.Y = HM(Y); g.ldy = Cout;
    return uv_launch_gemm(g, 1, S(s));
}
int univst_conv_nhwc_tapinner(const void* X1, const void* X2, int C1, int C2, int imgs, int Hs, int Ws, int up, int stride,
                              const void* W, const void* bias, const void* rowbias, int rows_per_rb, const void* R, void* Y, int Cout,
                              void* s) {
    UV_REQUIRE(X1 && W && Y, "conv: null argument");
    UV_REQUIRE((X2 != nullptr) == (C2 > 0), "conv: X2/C2 mismatch");
    GemmParams g;
    g.X = H(X1); g.X2 = H(X2); g.C1 = C1; g.C2 = C2; g.Hs = Hs; g.Ws = Ws; g.up = up ? 1 : 0; g.stride = stride; g.taps = 9; g.korder = 1;
    const int He = Hs << g.up, We = Ws << g.up;
    g.Ho = (He - 1) / stride + 1;
    g.Wo = (We - 1) / stride + 1;
    g.M = imgs * g.Ho * g.Wo; g.N = Cout; g.K = 9 * (C1 + C2);
    g.W = H(W); g.bias = H(bias); g.rowbias = H(rowbias); g.rows_per_rb = rows_per_rb > 0 ? rows_per_rb : 1;
    g.R = H(R); g.ldr = Cout; g.Y = HM(Y); g.ldy = Cout;
    return uv_launch_gemm(g, 1, S(s));
}
int univst_conv3x3_patch(const void* X1, const void* X2, int C1, int C2, int imgs, int Hs, int Ws, int upsample, const void* W32,
                         const void* bias, const void* rowbias, int rows_per_rb, const void* R, void* Y, int Cout, void* s) {
    UV_REQUIRE(X1 && W32 && Y, "conv3x3_patch: null argument");
    UV_REQUIRE((X2 != nullptr) == (C2 > 0), "conv3x3_patch: X2/C2 mismatch");
    GemmParams g;
    g.X = H(X1); g.X2 = H(X2); g.C1 = C1; g.C2 = C2; g.Hs = Hs; g.Ws = Ws; g.up = upsample ? 1 : 0; g.stride = 1; g.taps = 9;
    g.Ho = Hs << g.up; g.Wo = Ws << g.up;
    g.M = imgs * g.Ho * g.Wo; g.N = Cout; g.K = 9 * (C1 + C2);
    g.W = nullptr; g.W32 = H(W32);
    g.bias = H(bias); g.rowbias = H(rowbias); g.rows_per_rb = rows_per_rb > 0 ? rows_per_rb : 1;
    g.R = H(R); g.ldr = Cout; g.Y = HM(Y); g.ldy = Cout;
    return uv_launch_gemm(g, 1, S(s));
}
int univst_groupnorm_fold_linear(const void* X, int C, int64_t rows, int rows_per_stat, int groups, float eps, const void* gamma, const void* beta,
                                 const void* W, const void* bias, int N, void* W_sets, float* bias32, void* ws, void* s) {
    UV_REQUIRE(X && gamma && beta && W && W_sets && bias32 && ws && N > 0, "groupnorm_fold_linear: null argument");
    UvGnFold f;
    f.W = H(W); f.bias = H(bias); f.N = N; f.W_out = HM(W_sets); f.bias32 = bias32;
    return uv_launch_groupnorm(H(X), nullptr, C, 0, rows, rows_per_stat, groups, eps, H(gamma), H(beta), 0, nullptr, (float*)ws, S(s), nullptr, nullptr, nullptr, &f);
}
int univst_linear_sets(const void* X, int64_t ldx, const void* W_sets, const float* bias32, int rows_per_set, const void* R, int64_t ldr, void* Y, int64_t ldy,
                       int M, int N, int K, float* stats_out, void* s) {
    UV_REQUIRE(X && W_sets && bias32 && Y && rows_per_set > 0, "linear_sets: null argument");
    GemmParams g;
    g.X = H(X); g.ldx = ldx; g.W = H(W_sets); g.bias32 = bias32; g.w_rows_per_set = rows_per_set;
    g.R = H(R); g.ldr = ldr; g.Y = HM(Y); g.ldy = ldy; g.M = M; g.N = N; g.K = K; g.stats_out = stats_out;
    UV_REQUIRE(uv_linear_takes_big_direct(M, N, K, ldx), "linear_sets: M=%d N=%d K=%d is not a problem the direct 256x320 path takes (N %% 320 == 0, >= 150 tiles)", M, N, K);
    return uv_launch_gemm(g, 0, S(s));
}
int64_t univst_groupnorm_workspace_bytes(int64_t rows, int rows_per_stat, int groups) {
    if (rows_per_stat <= 0) return 0;
    return (int64_t)uv_groupnorm_workspace_floats((int)(rows / rows_per_stat), groups) * 4;
}
int univst_groupnorm_nhwc(const void* X1, const void* X2, int C1, int C2, int64_t rows, int rows_per_stat, int groups, float eps,
                          const void* gamma, const void* beta, int silu, void* Y, void* ws, void* s) {
    UV_REQUIRE(X1 && gamma && beta && Y && ws, "groupnorm: null argument");
    return uv_launch_groupnorm(H(X1), H(X2), C1, C2, rows, rows_per_stat, groups, eps, H(gamma), H(beta), silu, HM(Y), (float*)ws,
                               S(s));
}
int univst_layernorm(const void* X, void* Y, const void* gamma, const void* beta, int64_t rows, int C, float eps, void* s) {
    UV_REQUIRE(X && Y && gamma && beta, "layernorm: null argument");
    return uv_launch_layernorm(H(X), C, HM(Y), C, H(gamma), H(beta), rows, C, eps, S(s));
}
int univst_attention(const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, void* out, int64_t ldo,
                     const int32_t* src_idx, const int32_t* src_cnt, const float* src_logw, int nsrc, int BF, int Nq, int Nkv, int heads,
                     int d, int q_prescaled, void* s) {
    UV_REQUIRE(q && k && v && out && src_idx, "attention: null argument");
    AttnParams a;
    a.src_cnt = src_cnt;
    a.src_logw = src_logw;
    a.q = H(q); a.k = H(k); a.v = H(v); a.o = HM(out); a.ldq = ldq; a.ldkv = ldkv; a.ldo = ldo; a.src_idx = src_idx; a.nsrc = nsrc;
    a.BF = BF; a.Nq = Nq; a.Nkv = Nkv; a.heads = heads; a.d = d;
    a.scale_log2e = 1.4426950408889634f / sqrtf((float)d);
    a.q_prescaled = q_prescaled != 0;
    return uv_launch_attention(a, S(s));
}
int univst_attention_phase(const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, void* out, int64_t ldo,
                           const int32_t* src_idx, const int32_t* src_cnt, const float* src_logw, int nsrc, int BF, int Nq, int Nkv, int heads,
                           int d, int q_prescaled, float* state_out, const float* state_in, void* s) {
    UV_REQUIRE(q && k && v && out && src_idx && src_cnt, "attention_phase: null argument (src_cnt is required: a phase may leave a frame without sources)");
    UV_REQUIRE((state_out != nullptr) != (state_in != nullptr), "attention_phase: exactly one of state_out (first phase) / state_in (second phase)");
    AttnParams a;
    a.src_cnt = src_cnt;
    a.src_logw = src_logw;
    a.q = H(q); a.k = H(k); a.v = H(v); a.o = HM(out); a.ldq = ldq; a.ldkv = ldkv; a.ldo = ldo; a.src_idx = src_idx; a.nsrc = nsrc;
    a.BF = BF; a.Nq = Nq; a.Nkv = Nkv; a.heads = heads; a.d = d;
    a.scale_log2e = 1.4426950408889634f / sqrtf((float)d);
    a.q_prescaled = q_prescaled != 0;
    a.state_out = state_out;
    a.state_in = state_in;
    return uv_launch_attention(a, S(s));
}
int univst_attention_adain_shift(void* qkv, int64_t ld, int F, int N, int C, float alpha, float beta, float gamma, void* ws, void* s) {
    UV_REQUIRE(qkv && ws, "adain_shift: null argument");
    float* w = (float*)ws;
    return uv_launch_adain_shift(HM(qkv), ld, F, N, C, w, w + (long)F * 2 * C, alpha, beta, gamma, S(s));
}
int univst_latent_adain(const void* cnt, const void* sty, void* out, int C, int F, int HW, void* s) {
    UV_REQUIRE(cnt && sty && out, "latent_adain: null argument");
    return uv_launch_latent_adain(H(cnt), H(sty), HM(out), C, F, HW, S(s));
}
int univst_latent_adain_stats(const void* cnt, float* stats, int C, int F, int HW, void* s) {
    UV_REQUIRE(cnt && stats, "latent_adain_stats: null argument");
    return uv_launch_latent_adain_stats(H(cnt), stats, C, F, HW, S(s));
}
int univst_latent_adain_apply(const void* cnt, const void* sty, const float* stats, int64_t n_total, void* out, int C, int F, int HW,
                              void* s) {
    UV_REQUIRE(cnt && sty && stats && out && n_total > 0, "latent_adain_apply: bad argument");
    return uv_launch_latent_adain_apply(H(cnt), H(sty), stats, n_total, HM(out), C, F, HW, S(s));
}
int univst_axpby(const void* x, const void* e, void* out, float cx, float ce, int64_t n, void* s) {
    UV_REQUIRE(x && e && out, "axpby: null argument");
    return uv_launch_axpby(H(x), H(e), HM(out), cx, ce, n, S(s));
}
int univst_mask_blend(const void* a, const void* b, const void* m, void* out, int C, int64_t FHW, void* s) {
    UV_REQUIRE(a && b && out, "mask_blend: null argument");
    return uv_launch_mask_blend(H(a), H(b), H(m), HM(out), C, FHW, S(s));
}
int univst_mask_resize(const uint8_t* mask, void* out, int F, int Hh, int W, int h, int w, void* s) {
    UV_REQUIRE(mask && out, "mask_resize: null argument");
    return uv_launch_mask_resize(mask, HM(out), F, Hh, W, h, w, S(s));
}
int univst_debug_tr16(float* out, void* s) { return uv_launch_tr16_probe(out, S(s)); }
int univst_debug_delay_us(double us, void* s) { return uv_launch_delay_us(us, S(s)); }
int univst_profile_enable(int on) {
    uv_prof_enable(on);
    return UV_OK;
}
int univst_profile_symbols(int cls, char* buf, int n) {
    UV_REQUIRE(buf && n > 0, "profile_symbols: null buffer");
    const std::string sy = uv_prof_symbols(cls);
    snprintf(buf, (size_t)n, "%s", sy.c_str());
    return UV_OK;
}
int univst_profile_collect(double* ms, int64_t* count, double* flops, double* bytes, int ncls) {
    UV_REQUIRE(ms && count && flops && bytes && ncls >= 1 && ncls <= 16, "profile_collect: bad arguments");
    long c[16];
    int rc = uv_prof_collect(ms, c, flops, bytes, ncls);
    for (int i = 0; i < ncls; ++i) count[i] = c[i];
    return rc;
}

int univst_profile_collect_aux(double* aux_bytes, int ncls) {
    UV_REQUIRE(aux_bytes && ncls >= 1 && ncls <= 16, "profile_collect_aux: bad arguments");
    uv_prof_aux(aux_bytes, ncls);
    return UV_OK;
}

int64_t univst_maskprop_workspace_bytes(int hw, int Nsrc, int C) { return uv_maskprop_workspace_bytes(hw, Nsrc, C); }
int univst_maskprop_frame(const float* feat_tar, const float* feat_src, const float* segs_src, float* segs_tar, int hw, int Nsrc,
                          int C, int ncls, float T, int topk, void* ws, void* s) {
    UV_REQUIRE(feat_tar && feat_src && segs_src && segs_tar && ws, "maskprop_frame: null argument");
    return uv_launch_maskprop_frame(feat_tar, feat_src, segs_src, segs_tar, hw, Nsrc, C, ncls, T, topk, ws, S(s));
}
int univst_maskprop_finalize(const float* segs, uint8_t* out, int ncls, int h, int w, int Hh, int W, void* ws, void* s) {
    UV_REQUIRE(segs && out && ws, "maskprop_finalize: null argument");
    return uv_launch_maskprop_finalize(segs, out, ncls, h, w, Hh, W, ws, S(s));
}
int univst_warp_accumulate(const uint8_t* key, const uint8_t* now, const float* fwd, const float* bwd, float* acc, int Hh, int W,
                           float thr, void* s) {
    UV_REQUIRE(key && now && fwd && bwd && acc, "warp_accumulate: null argument");
    return uv_launch_warp_accumulate(key, now, fwd, bwd, acc, Hh, W, thr, S(s));
}
int univst_warp_window_key(uint8_t* frames, const float* const* flows, int nn, int F, int Hh, int W, int key, int r, float thr, void* s) {
    UV_REQUIRE(frames, "warp_window_key: null argument");
    return uv_launch_warp_window_key(frames, flows, nn, F, Hh, W, key, r, thr, S(s));
}
int univst_latent_window_smooth(void* x0, const float* lflow, int C, int F, int h, int w, int r, float thr, void* s) {
    UV_REQUIRE(x0 && lflow, "latent_window_smooth: null argument");
    return uv_launch_latent_window_smooth(HM(x0), lflow, C, F, h, w, r, thr, S(s));
}
int univst_accumulate_u8(const uint8_t* f, float* acc, int64_t n, void* s) {
    UV_REQUIRE(f && acc, "accumulate_u8: null argument");
    return uv_launch_accumulate_u8(f, acc, n, S(s));
}
int univst_window_store(const float* acc, float weight, uint8_t* dst, int64_t n, void* s) {
    UV_REQUIRE(acc && dst && weight > 0.f, "window_store: bad argument");
    return uv_launch_window_store(acc, weight, dst, n, S(s));
}

}  // extern "C"
