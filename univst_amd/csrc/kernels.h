// Internal launcher interface shared by the kernel translation units, the UNet graph and the C ABI.
#pragma once
#include "common.h"
#include <string>

struct GemmParams {
    // Y[m][n] = epilogue(sum_k X[m][k] W[n][k])
    const half_t* X = nullptr;   // MODE 0: [M,K] (row stride ldx).  MODE 1: NHWC source 1 [imgs, Hs, Ws, C1]
    const half_t* X2 = nullptr;  // MODE 1: NHWC source 2 (virtual channel concat) or null
    const half_t* W = nullptr;   // [N,K], K contiguous (conv: K = tap*(C1+C2)+c)
    const half_t* W32 = nullptr; // optional second copy of a 3x3 conv weight in the k order [Cin/32][9][32] (LDS-patch kernel)
    half_t* Y = nullptr;
    long ldx = 0, ldy = 0;
    int M = 0, N = 0, K = 0;
    // MODE 1 geometry
    int C1 = 0, C2 = 0, Hs = 1, Ws = 1, up = 0, Ho = 1, Wo = 1, stride = 1, taps = 1;
    // kernel geometry (0 = the launcher's default: 3x3 with padding 1 for taps = 9, 1x1 for taps = 1).  Explicit forms (round 5, the VAE): the 3x1
    // frame conv of a Conv3d (3,1,1) on the geometry "image rows = frames, image columns = pixels" (taps = 3, tapw = 1, pady = 1, padx = 0) and the
    // encoder's stride-2 conv over the input padded at the bottom / right only (taps = 9, tapw = 3, pady = padx = 0)
    int tapw = 0, pady = 0, padx = 0;
    // korder 0: K index = tap*(C1+C2) + c.  korder 1 ("tap-inner", needs C1 % 64 == C2 % 64 == 0, taps == 9):
    // K index = (c/64)*576 + tap*64 + c%64 — the 9 taps of one 64-channel slab are consecutive k tiles, so the
    // activation rows a block re-reads for the shifted taps are still in its XCD's L2 (9x less beyond-L2 traffic).
    int korder = 0;
    // epilogue
    const half_t* bias = nullptr;      // [N]
    const half_t* rowbias = nullptr;   // [M / rows_per_rb, N]  (time-embedding add per branch)
    int rows_per_rb = 1;
    long ldrb = 0;                     // row stride of rowbias (0: N) — the UNet graph projects all 22 time embeddings in one launch
    const half_t* R = nullptr;         // residual [M, ldr]
    long ldr = 0;
    const half_t* bias2 = nullptr;     // second bias added after fp16 rounding (attn_temporal bias)
    int geglu = 0;                     // 1: W rows interleaved [16 x | 16 gate]; 2: the X-resident kernel's order (K = 320; uv_launch_geglu_xres_permute); writes N/2 columns x*gelu(gate)
    // MM-DiT epilogues (SD3 path; not with geglu / the LayerNorm fold): Y = R + gate[m / rows_per_gate] (.) act(acc + bias + rowbias)
    int act = 0;                       // 1: GELU(tanh)  (FeedForward activation_fn="gelu-approximate")
    const half_t* gate = nullptr;      // [M / rows_per_gate] rows of N gate values, ld_gate halfs apart (a chunk of the adaLN linear's output)
    long ld_gate = 0;
    int rows_per_gate = 1;
    int epi_lds = 0;                   // 256x320 kernel: transpose the tile through LDS for row-contiguous stores
    // split-K (small M*N, long K: the deep UNet levels, and every level of a frame shard): grid = tiles x splits, each block
    // reduces ktps k tiles into fp32 partials [splits][M][N]; a second kernel sums them in split order and applies the epilogue.
    float* partial = nullptr;          // caller-provided workspace (UNet arena) or null (the launcher then uses hipMallocAsync)
    size_t partial_bytes = 0;
    int splits = 1, ktps = 0;          // set by the launcher
    int tile_gn = 0, tile_gm = 0;      // 256x320 kernel: tile order in groups of tile_gm row tiles x tile_gn column tiles (0: column tile fastest); set by the launcher
    int issue_mode = 0;                // 256x320 kernel, A/B aid: how the next tile's DMA is spread over the current tile's MFMAs
    // LayerNorm folded into the linears around it (uv_linear_fold_producer_ok / uv_linear_fold_consumer_ok):
    //  producer: stats_out[m][N/160][2] <- (sum, sum of squares) of the stored fp16 outputs per 160-column slot;
    //  consumer: X holds the RAW rows, W = gamma (.) W, and the epilogue computes rstd*(acc - mean*ln_wsum[n]) + ln_bias[n]
    //            with (mean, rstd) over K from ln_stats[m][ln_slots][2]  (ln_bias already contains the linear's own bias).
    // GroupNorm statistics from the producer's epilogue (round 4; 256x320 kernels with the LDS epilogue, no GEGLU / split-K): for every
    // 16-row fragment m/16 and every SUB-GROUP of gn_gw = 10 channels (gn_G = N / 10 of them; 10 divides every group width of the UNet, also
    // those of a channel concat) the (sum, sum of squares) of the fp16 values AS STORED go to gn_out[sub-group][m/16][2]; uv_launch_groupnorm
    // sums fragments and sub-groups into its groups, so the statistics pass over the tensor (a third of a GroupNorm's traffic) disappears.  *gn_emitted is set to 1 when the launch took a
    // path that writes them (the caller falls back to the stand-alone pass otherwise).
    float* gn_out = nullptr;
    int gn_G = 0, gn_gw = 0;
    int* gn_emitted = nullptr;
    // per-row-range WEIGHT SETS (round 5: the transformer's per-frame GroupNorm folded into proj_in, uv_launch_groupnorm's `fold`): rows
    // [s * w_rows_per_set, (s + 1) * w_rows_per_set) use W + s * N * K and, in place of `bias`, bias32 + s * N (fp32: it carries sum_k W[n][k] *
    // (beta_k - mean * rstd * gamma_k), which cancels against the scaled activations when a group's mean is many standard deviations).  Direct
    // 256x320 linears whose tile height divides w_rows_per_set only.
    int w_rows_per_set = 0;
    const float* bias32 = nullptr;
    float* stats_out = nullptr;
    const float* ln_stats = nullptr;
    int ln_slots = 0;
    float ln_eps = 0.f;
    const float* ln_wsum = nullptr;    // [N] fp32: sum_k W'[n][k]
    const float* ln_bias = nullptr;    // [N] fp32: bias[n] + sum_k beta[k] W[n][k]
};

struct AttnParams {
    const half_t* q = nullptr;   // row (bf*Nq + i), head h at column h*d
    const half_t* k = nullptr;   // row (src*Nkv + j)
    const half_t* v = nullptr;
    half_t* o = nullptr;
    long ldq = 0, ldkv = 0, ldo = 0;
    const int* src_idx = nullptr;   // [BF][nsrc] source frame (row-block) of every key segment
    const int* src_cnt = nullptr;   // optional [BF]: only the first src_cnt[bf] sources are used
    const float* src_logw = nullptr;  // optional [BF][nsrc]: log2 multiplicity of a source (duplicate frames merged: softmax over a
                                      // key set that contains frame X m times == adding log2(m) to X's scores)
    int nsrc = 1, BF = 0, Nq = 0, Nkv = 0, heads = 0, d = 0;
    float scale_log2e = 0.f;
    int q_prescaled = 0;            // q rows already carry the factor scale_log2e (the UNet graph folds it into to_q for head_dim 40)
    int order = 1;                  // d = 40 pipelined kernel: 1 = head-major block order (L2 reuse of shared key frames), 0 = frame-major
    // one EXTRA key segment of its own length behind the equal-length sources (the text tokens of SD3's joint attention,
    // video_diffusion_sd3/pnp_utils.py:80-98: keys = [first | prev | cur] image tokens ++ text tokens): rows x_idx[bf] * Nkv_x ..
    // of kx / vx, head h at column h*d, no multiplicity.  Served by the generic kernel (attn_body) only.
    const half_t* kx = nullptr;
    const half_t* vx = nullptr;
    const int* x_idx = nullptr;     // [BF] row block of the extra segment for every query frame
    long ldkv_x = 0;
    int Nkv_x = 0;
    // TWO-PHASE attention (round 6; the frame shard consumes its LOCAL key frames while the halo frames are still on the wire): online softmax is
    // order independent, so the key set of a query may be split over two launches.  Phase 1 (state_out set) writes its normalised rows to o as usual
    // and leaves per (frame, head, query) the pair (m, l) = (reference of the exponentials in log2 units, denominator w.r.t. it) in
    // state_out[((bf * heads + h) * Nq + i) * 2]; a frame with src_cnt == 0 gets zero rows and l = 0.  Phase 2 (state_in set) runs over the remaining
    // sources and MERGES: o <- (o1 * l1 * 2^(m1 - m) + acc2 * 2^(m2 - m)) / (l1 * 2^(m1 - m) + l2 * 2^(m2 - m)), m = max(m1, m2) — exact up to the fp16
    // rounding of o1; frames with src_cnt == 0 keep their phase-1 rows.  Served by attn_body and attn_pp40_kernel (the dispatcher keeps such launches there).
    float* state_out = nullptr;
    const float* state_in = nullptr;
};

// The text cross-attention of a transformer block as ONE launch (fused.hip: attention.py:321-327 = norm2 -> attn2 -> + residual)
struct Attn2Params {
    const half_t* X = nullptr;          // [M, ldx] input rows: RAW rows when ln_stats is set (LayerNorm folded), else already normalised
    long ldx = 0;
    int M = 0;
    const float* ln_stats = nullptr;    // [M][ln_slots][2] (sum, sumsq) per 160-column slot from the producer of X (GemmParams::stats_out), or null
    int ln_slots = 0;
    float ln_eps = 1e-5f;
    const float* ln_wsum = nullptr;     // [C] fp32 (see GemmParams)
    const float* ln_bias = nullptr;     // [C] fp32
    const half_t* Wq_f = nullptr;       // to_q weight (gamma-folded when ln_stats) in MFMA operand order (uv_launch_frag_pack)
    const half_t* kvf = nullptr;        // text K | V of every (branch, head) in operand order (uv_launch_kv_frag_pack)
    int rows_per_branch = 0, heads = 0, Nkv = 0;
    int q_prescaled = 0;                // the to_q weight carries log2(e)/sqrt(d)
    float scale_log2e = 0.f;
    const half_t* Wo_f = nullptr;       // to_out weight in operand order
    const half_t* bias_o = nullptr;     // [C] or null
    const half_t* R = nullptr;          // residual rows [M, ldr]
    long ldr = 0;
    half_t* Y = nullptr;
    long ldy = 0;
    float* stats_out = nullptr;         // [M][C/160][2] row statistics of Y for a following folded LayerNorm, or null
    // the self-attention's out projection fused in front (attention.py:316-319): X = attention output rows, H2 = X Wp^T + bias_p + Rp is the block's input
    // AND residual and never reaches memory; the LayerNorm statistics are taken inside (ln_stats null, ln_wsum / ln_bias set)
    const half_t* Wp_f = nullptr;       // attn1.to_out weight in operand order, or null: no fused out projection
    const half_t* bias_p = nullptr;
    const half_t* Rp = nullptr;         // its residual rows [M, ldrp]
    long ldrp = 0;
#ifdef UV_A2_TRACE
    long long* trace = nullptr;         // tools/probes/attn2_probe.hip: [blocks][4 waves][8] cycle counter at the phase boundaries
#endif
};
bool uv_attn2_fused_ok(int C, int heads, int rows_per_branch, int Nkv);
long uv_attn2_kvf_halfs(int B, int heads, int D);
int uv_launch_frag_pack(const half_t* W, half_t* out, int N, int K, hipStream_t s);
int uv_launch_kv_frag_pack(const half_t* kv, half_t* out, int B, int T, int C, int heads, hipStream_t s);
int uv_launch_attn2_fused(const Attn2Params& p, int C, hipStream_t s);

int uv_launch_gemm(const GemmParams& p, int mode, hipStream_t stream);
bool uv_linear_takes_big_direct(long M, int N, int K, long ldx = 0);
// LayerNorm fold around a plain linear: may it emit the row statistics of its output / apply those of its input?  (256x320 direct path,
// or the 128-wide path when that runs the problem without split-K; the GEGLU consumer is 256x320 only)
bool uv_linear_fold_producer_ok(long M, int N, int K);
bool uv_linear_fold_consumer_ok(long M, int N, int K, bool geglu);
// GEGLU projection with K = 320 on the X-resident kernel (GemmParams::geglu = 2): shape test and the weight / bias row permutation it needs
bool uv_geglu_xres_ok(int N, int K, long M = 0);
int uv_launch_geglu_xres_permute(const half_t* in, half_t* out, int rows, int cols, hipStream_t stream);
constexpr size_t UV_SPLITK_WS_BYTES = 128u << 20;  // fp32 partials [splits][M][N]: 8 splits of the 8x8-level convs (3072 x 1280)
int uv_launch_linear_small(const half_t* x, const half_t* W, const half_t* b, half_t* y, int M, int N, int K, int silu_in,
                           hipStream_t stream);
struct UvGnComm {      // cross-rank reduction hook of the 5-D GroupNorm (frame sharding)
    int world = 1;
    float* red = nullptr;                                            // device [S*G*2] inside the comm workspace
    long byte_off = 0;                                               // its byte offset in that workspace
    int (*allreduce)(void* user, int64_t byte_off, int count_f32) = nullptr;
    void* user = nullptr;
};
int uv_groupnorm_workspace_floats(int S, int G);
// pre_part / pre_part2: statistics already emitted by the producers of s1 / s2 (GemmParams::gn_out, [C/10][rows/16][2] each: 10-channel
// sub-groups): the partial-sum pass over the tensor(s) is skipped when every source has them
// fold (round 5): instead of applying the normalisation, fold it into the linear that consumes the tensor — per stat unit s (a frame) the weight
// set W_s[n][k] = fp16(W[n][k] * gamma_k * rstd_{s,g(k)}) and the fp32 bias b_s[n] = bias[n] + sum_k W[n][k] * (beta_k - mean_{s,g(k)} * rstd * gamma_k):
// the linear then runs on the RAW tensor (GemmParams::w_rows_per_set / bias32) and the apply pass (read + write of the tensor) disappears.  `out` unused.
struct UvGnFold {
    const half_t* W = nullptr;       // [N][C] the consuming linear's weight
    const half_t* bias = nullptr;    // [N] or null
    int N = 0;
    half_t* W_out = nullptr;         // [S][N][C]
    float* bias32 = nullptr;         // [S][N]
};
int uv_launch_groupnorm(const half_t* s1, const half_t* s2, int C1, int C2, long rows, int rows_per_stat, int G, float eps,
                        const half_t* gamma, const half_t* beta, int silu, half_t* out, float* part, hipStream_t stream,
                        const UvGnComm* comm = nullptr, const float* pre_part = nullptr, const float* pre_part2 = nullptr, const UvGnFold* fold = nullptr);
int uv_launch_layernorm(const half_t* x, long ldx, half_t* y, long ldy, const half_t* gamma, const half_t* beta, long rows,
                        int C, float eps, hipStream_t stream);
int uv_launch_attention(const AttnParams& p, hipStream_t stream);
int uv_launch_tr16_probe(float* out, hipStream_t stream);
int uv_launch_colstats(const half_t* x, long ld, int F, int N, int ncols, float* mean, float* stdv, hipStream_t stream);
int uv_launch_adain_shift(half_t* qkv, long ld, int F, int N, int C, float* mean, float* stdv, float alpha, float beta,
                          float gamma, hipStream_t stream);
int uv_launch_latent_adain(const half_t* cnt, const half_t* sty, half_t* out, int Cl, int F, int HW, hipStream_t stream);
int uv_launch_latent_adain_stats(const half_t* cnt, float* st, int Cl, int F, int HW, hipStream_t stream);
int uv_launch_latent_adain_apply(const half_t* cnt, const half_t* sty, const float* st, long n_total, half_t* out, int Cl, int F, int HW,
                                 hipStream_t stream);
int uv_launch_rows_pack(const half_t* src, long ld, int col0, int width, int N, int B, int frames_per_branch, int frame, half_t* dst, hipStream_t s);
int uv_launch_ncfhw_to_nhwc(const half_t* x, half_t* y, int B, int Cl, int F, int HW, int CP, hipStream_t s);
int uv_launch_nhwc_to_ncfhw(const half_t* x, int ldx, half_t* y, int B, int Cl, int F, int HW, hipStream_t s);
int uv_launch_timestep_embed(float t, half_t* out, int B, int dim, int flip, float shift, hipStream_t s);
int uv_launch_axpby(const half_t* x, const half_t* e, half_t* out, float cx, float ce, long n, hipStream_t s);
int uv_launch_mask_blend(const half_t* a, const half_t* b, const half_t* m, half_t* out, int Cl, long FHW, hipStream_t s);
int uv_launch_mask_resize(const uint8_t* mask, half_t* out, int F, int H, int W, int h, int w, hipStream_t s);
int uv_launch_add_bias_rows(half_t* x, const half_t* b, long rows, int C, hipStream_t s);
int64_t uv_maskprop_workspace_bytes(int hw, int Nsrc, int C);
int uv_launch_maskprop_frame(const float* feat_tar, const float* feat_src, const float* segs_src, float* segs_tar, int hw,
                             int Nsrc, int C, int ncls, float T, int topk, void* ws, hipStream_t s);
int uv_launch_maskprop_finalize(const float* segs, uint8_t* out, int ncls, int h, int w, int H, int W, void* ws, hipStream_t s);
int uv_launch_warp_accumulate(const uint8_t* key, const uint8_t* now, const float* fwd, const float* bwd, float* acc, int H, int W,
                              float thr, hipStream_t s);
int uv_launch_warp_window_key(uint8_t* est, const float* const* flows, int nn, int F, int H, int W, int key, int r, float thr, hipStream_t s);
int uv_launch_latent_window_smooth(half_t* x0, const float* lflow, int C, int F, int h, int w, int r, float thr, hipStream_t s);
int uv_launch_delay_us(double us, hipStream_t s);
int uv_launch_accumulate_u8(const uint8_t* f, float* acc, long n, hipStream_t s);
int uv_launch_window_store(const float* acc, float weight, uint8_t* dst, long n, hipStream_t s);

// ---- optional per-class HIP-event profiling (prof.hip)
// one class per kernel SYMBOL that matters, so bench.py's average launch duration is comparable with the
// rocprofv3 --stats row of the same name
enum {
    UV_CLS_GEMM_BIG = 0,    // gemm_big_kernel<0> and geglu_xres_kernel (the K = 320 GEGLU projection)
    UV_CLS_CONV_BIG = 1,    // gemm_big_kernel<1>
    UV_CLS_GEMM = 2,        // gemm_kernel<*,0,...>
    UV_CLS_CONV = 3,        // gemm_kernel<*,1,...>
    UV_CLS_ATTN_D40 = 4,    // attn_pp40_kernel<true>: head_dim 40 self-attention over >= 2048 queries (the 64x64 level)
    UV_CLS_ATTN_D80 = 5,    // attn_kernel*<96,5,*>
    UV_CLS_ATTN_OTHER = 6,  // remaining attention instantiations
    UV_CLS_GROUPNORM = 7,
    UV_CLS_LAYERNORM = 8,
    UV_CLS_ADAIN = 9,
    UV_CLS_ATTN_TEXT = 10,  // the 77-key text cross-attention launches (any head_dim): reported apart from the self-attention
    UV_CLS_CONV_PATCH = 11, // conv_patch_kernel<*>: 3x3 / stride-1 convs from an input patch kept in LDS
    UV_CLS_ATTN2_FUSED = 12, // attn2_fused_kernel: q projection + 77-key text attention + out projection + residual in one launch
    UV_NCLS = 13
};
void uv_prof_enable(int on);
bool uv_prof_on();
void uv_prof_begin(int cls, double flops, double bytes, hipStream_t s, const char* sym = nullptr, double aux_bytes = 0.0);     // sym: the kernel symbol about to be launched (template arguments without spaces)
void uv_prof_end(hipStream_t s);
void uv_prof_aux(double* aux, int ncls);          // per class: the `aux_bytes` sums of the records the LAST uv_prof_collect returned
int uv_prof_collect(double* ms, long* count, double* flops, double* bytes, int ncls);
std::string uv_prof_symbols(int cls);           // ';'-joined symbols launched in that class since profiling was switched on
