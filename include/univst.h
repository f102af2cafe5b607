/* univst.h — C ABI of libunivst_hip.so: the MI355X (gfx950) implementation of the UniVST SD-v1.5
 * denoising hot path.  Plain pointers and sizes only; no torch types.  All pointers are DEVICE pointers
 * unless stated otherwise; all tensors are fp16 (IEEE binary16) unless stated otherwise; all work is
 * enqueued on the caller-supplied hipStream_t (passed as void*) and the call returns without syncing.
 *
 * Every entry returns 0 on success or a negative code (UNIVST_ERR_*); the message is available through
 * univst_last_error() (thread-local).  Handles are not re-entrant: one call at a time per handle.
 *
 * The reference (QuanjianSong/UniVST) is pure Python with no native layer, so there is no existing FFI to
 * mimic; each entry names the reference interface (file:line under the reference checkout) it replaces.
 * The Python binding a maintainer adds is univst_amd/_native.py (ctypes); see INTEGRATION.md.
 */
#ifndef UNIVST_H
#define UNIVST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define UNIVST_OK 0
#define UNIVST_ERR_ARG (-1)
#define UNIVST_ERR_HIP (-2)
#define UNIVST_ERR_UNSUPPORTED (-3)
#define UNIVST_ERR_STATE (-4)

#define UNIVST_F16 0
#define UNIVST_F32 1

const char* univst_last_error(void);
/* Bumped whenever an existing entry changes its signature or meaning (2: univst_sd3_joint_attention takes (shift, beta) instead of the
 * window parameters, round 4).  A binding checks it at load time (univst_amd/_native.py does): a caller built against an older header would
 * otherwise pass silently misinterpreted arguments. */
#define UNIVST_ABI_VERSION 3
int univst_abi_version(void);

/* ------------------------------------------------------------------ UNet handle
 * Replaces backbones/video_diffusion_sd/models/unet_3d_condition.py:49-230 (ctor), :306-443 (forward),
 * :445-509 (from_2d_model / load_2d_state_dict). */
typedef struct univst_unet univst_unet;

typedef struct {
    int in_channels, out_channels;
    int block_out_channels[4];
    int layers_per_block;
    int cross_attention_dim;
    int attention_heads[4];     /* SD config key attention_head_dim per down level, used as HEAD COUNT (unet_3d_blocks.py:269-271);
                                   SD-v1.x: 8,8,8,8 (head_dim = C/8); SD-v2.x: 5,10,20,20 (head_dim 64) */
    int norm_num_groups;
    float norm_eps;
    int flip_sin_to_cos;
    float freq_shift;
} univst_unet_cfg;

/* PnP state = what register_spatial_attention_pnp / register_time (pnp_utils.py:7-111) poke into attn1. */
typedef struct {
    int registered;             /* 1: the 8 decoder attn1 layers use the PnP forward ([-1,'first'] K/V sources) */
    int idx;                    /* step index (register_time) */
    float eta1, eta2;           /* window: eta1 <= idx <= eta2*50 */
    float alpha, gamma;         /* 0.65, 3.0 (pnp_utils.py:49-51) */
} univst_pnp;

int univst_unet_create(const univst_unet_cfg* cfg, univst_unet** out);
int univst_unet_destroy(univst_unet* h);
/* key = reference state-dict name (incl. *_temporal*); data is copied (D2D) and converted to fp16 */
int univst_unet_load_tensor(univst_unet* h, const char* key, const void* dev_ptr, int dtype, const int64_t* shape,
                            int ndim, void* stream);
/* builds the derived weight layouts ([Cout][ky][kx][Cin] convs, fused QKV / KV, GEGLU-interleaved FF) and
 * inspects the *_temporal* layers on the device: units still at their identity initialisation (dirac conv1d, zero
 * attn_temporal.to_out: every 2-D-initialised checkpoint) are skipped exactly; TRAINED units (a fine-tuned 3-D checkpoint) get their
 * derived layouts and run (resnet.py:70-80 as a 3-tap conv over frames through the conv kernels, attention.py:336-346 in a frame
 * attention kernel).  Trained temporal layers and frame sharding exclude each other (forward returns UNIVST_ERR_ARG). */
int univst_unet_finalize(univst_unet* h, void* stream);
/* (re)allocates the activation arena for this geometry; forward() calls it implicitly on first use */
int univst_unet_reserve(univst_unet* h, int B, int F, int H, int W);
/* sample [B,Cin,F,H,W], text [B,77,Dtxt] -> eps [B,Cout,F,H,W].  feat_out (may be NULL): receives the output
 * of up_blocks[ft_index] for batch element 0 as [F,H',W',C] (unet_3d_condition.py:430-436). */
int univst_unet_forward(univst_unet* h, const void* sample, float timestep, const void* text, int B, int F, int H,
                        int W, int text_len, const univst_pnp* pnp, void* eps_out, void* feat_out, int ft_index,
                        void* stream);
/* frame-sharded multi-GPU: this rank holds frames [rank*F, (rank+1)*F) of a clip of world*F frames
 * for ALL branches.  comm_ws is a caller-owned device workspace (>= 64 KiB + 4 x the largest frame pack); the callbacks are invoked on the
 * host while forward() enqueues work and must enqueue the collective on the SAME stream forward() was given (RCCL through
 * torch.distributed on the Python side):
 *   allreduce(user, byte_off, n)         : in-place SUM over ranks of n fp32 at comm_ws + byte_off  (5-D GroupNorm)
 *   kv_exchange(user, off_send_last, off_first, off_recv_prev, off_recv_first, nbytes):
 *        rank 0 broadcasts [off_first, +nbytes) (others receive it at off_recv_first); every rank r < world-1
 *        sends [off_send_last, +nbytes) to r+1, every rank r > 0 receives it at off_recv_prev  (sparse-causal attention, attention.py:384-413).
 * ABI 3 (round 6): a pack is the transformer block's HIDDEN rows of the boundary frame of every branch, [B, N, C] fp16 (the input of norm1 ->
 * to_k | to_v; until ABI 2: the K | V rows, [B, N, 2C]); the receiving rank projects (and, inside the PnP window, shifts) the two halo frames
 * itself and runs the attention in two phases (univst_attention_phase).  The callback only moves bytes and is invoked between the phases. */
typedef int (*univst_allreduce_fn)(void* user, int64_t byte_off, int count_f32);
typedef int (*univst_kv_exchange_fn)(void* user, int64_t off_send_last, int64_t off_first, int64_t off_recv_prev,
                                     int64_t off_recv_first, int64_t nbytes);
int univst_unet_set_comm(univst_unet* h, int rank, int world, void* comm_ws, int64_t comm_ws_bytes,
                         univst_allreduce_fn ar, univst_kv_exchange_fn kv, void* user);

/* The library's own frame-shard communicator (csrc/comm.hip): one process per GPU of ONE node, peers' regions mapped
 * through HIP IPC, all three couplings (GroupNorm statistics all-reduce, K/V halo + first-frame broadcast, latent_adain statistics)
 * as device-side peer writes + flags on the caller's stream — no host callbacks, so a forward() is a pure stream of kernels.
 * Replaces what the reference would need for `attention.py:384-413` / the 5-D GroupNorms of `resnet.py:338,369` across GPUs (the
 * reference itself is single-GPU).  Bring-up sequence on every rank:
 *   univst_comm_create(rank, world, ws_bytes, &c)      ws_bytes >= 64 KiB + 6 x the largest frame pack (B*N*C fp16 of hidden rows; B*N*2C keeps room for the SD3 K | V packs)
 *   univst_comm_export(c, handle)                      univst_comm_handle_bytes() bytes, to be all-gathered out of band
 *   univst_comm_connect(c, all_handles)                rank-major array of every rank's handle
 *   univst_unet_set_comm_native(unet, c)               the UNet graph now uses it instead of callbacks
 * Every wait is bounded; a rank that gives up records a code (univst_comm_status) and the next call returns UNIVST_ERR_STATE. */
typedef struct univst_comm univst_comm;
int univst_comm_create(int rank, int world, int64_t ws_bytes, univst_comm** out);
int univst_comm_handle_bytes(void);
int univst_comm_export(univst_comm* c, void* handle_out);
int univst_comm_connect(univst_comm* c, const void* handles);
/* ranks that are host threads of ONE process (tests on a 1-GPU box): `all` = the `world` communicators in rank order */
int univst_comm_connect_local(univst_comm* c, univst_comm* const* all);
/* ONE rank of a `world`-rank job alone on one GPU (bench.py --emulate-rank r/w --comm-emulated; measurement aid, results are meaningless): every peer
 * is this rank's own region; an exchange's post becomes a delay of latency + (packs on this rank's busiest link) x bytes / link rate followed by a
 * raise of its own inbox flags, on the stream the multicast would run on; an all-reduce runs over this rank alone after `latency_us`.  Kernels, streams
 * and flag waits are the production ones.  univst_comm_query("emu_wire_us"): the modelled time issued since the last query. */
int univst_comm_connect_emulated(univst_comm* c, double link_gbps, double latency_us);
int univst_comm_query(univst_comm* c, const char* name, double* out);
int univst_comm_destroy(univst_comm* c);
/* in-place SUM over ranks of n <= 1024 fp32 in device memory, identical bits on every rank (slots summed in rank order) */
int univst_comm_allreduce_f32(univst_comm* c, void* buf, int n, void* stream);
int univst_comm_status(univst_comm* c);
int univst_unet_set_comm_native(univst_unet* h, univst_comm* comm);

/* tuning switches of a handle (not part of the reference's surface; tests use them for A/B runs).
 *   "ln_fold" (default 2 since round 4, env UNIVST_LN_FOLD): fold the transformer blocks' LayerNorms (attention.py:311,321,329) into the
 *             linears around them instead of launching them separately: 0 none, 1 norm1 + norm2, 2 also norm3 (-> GEGLU).
 *   "gn_producer" (default 1, env UNIVST_GN_PRODUCER): GroupNorm statistics (resnet.py:338,369, attention.py:121) are taken from the epilogue
 *             of the conv / linear that writes the tensor instead of a separate pass over it, wherever that kernel is the 256x320 tile
 *             (0: always the stand-alone pass; results differ by fp32 summation order only).
 *   "chain_bands" (default 1 = off, env UNIVST_CHAIN_BANDS): the row-local chain behind a transformer block's self-attention
 *             (to_out -> attn2 -> to_out -> GEGLU feed-forward, attention.py:316-329) runs band by band over whole frames so that a band's
 *             intermediates stay in the 256 MB Infinity Cache: 0 = auto (bands of >= 65 536 rows), n = n bands.  Same results up to fp32
 *             summation order (a band may take another tile than the full tensor).  An
 *             experiment kept as a switch: -13 % on the isolated chain (tools/bench_mall_bands.py), +0.4 ms per step in the graph.
 *   "gn_fold" (default 1, env UNIVST_GN_FOLD): the per-frame GroupNorm in front of a transformer block (attention.py:121) is folded into proj_in as per-frame
 *             weight sets + an fp32 bias where the copies are cheap against the apply pass they replace (the 64x64 level); 0: always the apply pass.
 *   "attn2_fused" (default 2, env UNIVST_ATTN2_FUSED=0 disables it library-wide): the text cross-attention of a transformer block (attention.py:321-327)
 *             as one launch (univst_attn2_fused) where the level's shape is served, instead of q projection + attention + out projection; 2 (default): with the
 *             self-attention's out projection + residual in front of it (univst_attn12_fused), 1: attn2 alone, 0: off.
 *   "kv_overlap" (default 1, env UNIVST_KV_OVERLAP; round 6): the frame shard's K/V exchange of a transformer block (attention.py:384-413 across GPUs)
 *             is issued on a forked stream as soon as proj_in has written the boundary frames' hidden rows and joined in front of the halo phase of
 *             the two-phase attention; 0: issued on the forward's own stream (serial; A/B aid).  Results are identical.
 *   "emu_wire_gbps" / "emu_wire_lat_us" (default 0 = off / 3; bench.py --emulate-wire): with a callback communicator that moves nothing, every
 *             exchange occupies the forked stream for latency + (packs on this rank's busiest link) x pack bytes / rate — a 1-GPU box's stand-in for
 *             the xGMI transfer; the accumulated time is univst_unet_query("emu_wire_us"). */
int univst_unet_set_option(univst_unet* h, const char* name, int value);
/* read-outs of a handle: "emu_wire_us" (modelled wire time issued since the last query; reading resets it), "arena_high_water" (bytes). */
int univst_unet_query(univst_unet* h, const char* name, double* out);

/* ------------------------------------------------------------------ temporal VAE handle
 * The VAE behind the pipeline's decode / encode call sites (stable_diffusion.py:369-394 decode_latents, :793-818 get_images_from_latents,
 * :820-834 get_latent_image, ddim_inversion.py:28-31,52-55): diffusers' AutoencoderKLTemporalDecoder (the SVD VAE the reference's run_*_sd.py
 * load, src/sd/run_video_style_transfer_sd.py:36) as one graph of the library's kernels per call.  THIRD-PARTY network, restated from its published
 * definition with that class's state-dict keys (csrc/vae.hip); parity unpinned by the reference. */
typedef struct univst_vae univst_vae;
typedef struct {
    int in_channels, out_channels, latent_channels;      /* 3, 3, 4 */
    int block_out_channels[4];                           /* SVD: 128, 256, 512, 512 (multiples of norm_num_groups and of 8) */
    int layers_per_block;                                /* 2 */
    int norm_num_groups;                                 /* 32 */
} univst_vae_cfg;
int univst_vae_create(const univst_vae_cfg* cfg, univst_vae** out);
int univst_vae_destroy(univst_vae* h);
/* key = diffusers state-dict name ("decoder.up_blocks.0.resnets.1.temporal_res_block.conv1.weight", "quant_conv.bias", ...) */
int univst_vae_load_tensor(univst_vae* h, const char* key, const void* dev_ptr, int dtype, const int64_t* shape, int ndim, void* stream);
int univst_vae_finalize(univst_vae* h, void* stream);
/* AutoencoderKLTemporalDecoder.decode(z, num_frames).sample: z [imgs, latent, h, w] (imgs = clips x num_frames, already divided by the scaling
 * factor) -> out [imgs, out_channels, 8h, 8w]; the temporal layers couple the num_frames frames of a clip */
int univst_vae_decode(univst_vae* h, const void* z, int64_t imgs, int num_frames, int lat_h, int lat_w, void* out, void* stream);
/* AutoencoderKLTemporalDecoder.encode(x).latent_dist.parameters: x [imgs, in_channels, H, W] in [-1, 1] -> moments [imgs, 2*latent, H/8, W/8]
 * (mean | logvar; sampling stays with the caller, which owns the RNG) */
int univst_vae_encode(univst_vae* h, const void* x, int64_t imgs, int H, int W, void* moments, void* stream);

/* ------------------------------------------------------------------ stand-alone operators (also used by tests) */
/* Y[M,N] = X[M,K] W[N,K]^T + bias + residual; geglu != 0: the diffusers GEGLU projection (FeedForward net[0], attention.py:241) — writes the
 * N/2 columns x * gelu(gate), W / bias rows pre-interleaved: geglu = 1 in blocks of [16 x rows | 16 gate rows] (any K), geglu = 2 in the
 * order of the X-resident kernel for K = 320 (N % 256 == 0, no residual; univst_geglu_xres_permute makes it from the [x | gate] weight).
 * Replaces torch Linear / 1x1 conv call sites attention.py:123,141,375-377,425. */
int univst_linear(const void* X, int64_t ldx, const void* W, const void* bias, const void* residual, int64_t ldr,
                  void* Y, int64_t ldy, int M, int N, int K, int geglu, void* stream);
/* Row permutation of a GEGLU projection's weight [rows][cols] (or bias / per-row fp16 vector: cols = 1) from the checkpoint's
 * [x rows | gate rows] order into the geglu = 2 order: row nt*256 + wn*64 + i*16 + g*4 + r <- x row (r < 2) or gate row (r >= 2) of hidden
 * column nt*128 + wn*32 + i*8 + g*2 + (r & 1).  rows % 256 == 0. */
int univst_geglu_xres_permute(const void* in, void* out, int rows, int cols, void* stream);
/* The linear with the MM-DiT epilogues of the SD3 path (diffusers JointTransformerBlock / FeedForward, third-party):
 *   Y = residual + gate[m / rows_per_gate] (.) act(X W^T + bias)
 * act: UNIVST_ACT_NONE or UNIVST_ACT_GELU_TANH (defined below); gate (may be NULL): rows of N halfs, ld_gate apart, 16-byte aligned
 * (a chunk of the adaLN linear's [B, 6C] output: rows_per_gate = tokens per frame); residual may be NULL. */
int univst_linear_gated(const void* X, int64_t ldx, const void* W, const void* bias, const void* residual, int64_t ldr, void* Y, int64_t ldy,
                        int M, int N, int K, int act, const void* gate, int64_t ld_gate, int rows_per_gate, void* stream);
/* The same linear with a LayerNorm folded into it and / or row statistics emitted for the next one (what the UNet graph does with
 * norm1/2/3 of a transformer block, attention.py:311-329).  For problems the direct 256x320 tile takes (N % 320 == 0, at least
 * 150 tiles) and, without GEGLU, for problems the 128-wide tile runs without split-K (N % 160 == 0 for stats_out); else
 * UNIVST_ERR_ARG — a split-K problem has its epilogue in the reduction kernel.
 *   stats_out (may be NULL): fp32 [M][N/160][2] <- (sum, sum of squares) of the stored fp16 outputs per 160-column slot.
 *   ln_stats  (may be NULL): fp32 [M][K/160][2] written by the linear that produced X.  Then X holds the RAW rows, W must be
 *             fp16(gamma[k] * W[n][k]), ln_wsum[n] = sum_k of that, ln_bias[n] = bias[n] + sum_k beta[k] W[n][k] (fp32), bias NULL,
 *             and Y = rstd * (X W^T - mean * ln_wsum) + ln_bias  ==  LayerNorm(X) W_orig^T + bias  (+ residual, GEGLU as usual). */
int univst_linear_ln(const void* X, int64_t ldx, const void* W, const void* bias, const void* residual, int64_t ldr,
                     void* Y, int64_t ldy, int M, int N, int K, int geglu, const float* ln_stats, float ln_eps,
                     const float* ln_wsum, const float* ln_bias, float* stats_out, void* stream);
/* The text cross-attention of a transformer block as ONE launch (attention.py:321-327: norm2 -> attn2 -> + hidden_states; the attention itself
 * is diffusers' Attention with 77 text keys, third-party):
 *     Y = to_out( softmax( (LN(X) Wq^T)(text Wk^T)^T / sqrt(d) ) (text Wv^T) ) + bias_o + residual
 * replacing univst_linear_ln (q) + univst_attention (one 77-key source) + univst_linear (out + residual) and the HBM round trips of Q and O.
 *   X [M, ldx] input rows; with ln_stats (fp32 [M][C/160][2], written by the linear that produced X: univst_linear_ln's stats_out) X holds the
 *       RAW rows and Wq_frag is made from fp16(gamma[k] * Wq[n][k]), ln_wsum / ln_bias as in univst_linear_ln; else X is already normalised.
 *   Wq_frag, Wo_frag: the [C, C] weights in MFMA operand order (univst_frag_pack); q_prescaled != 0: Wq already carries log2(e)/sqrt(d).
 *   kv [B*T, 2C]: K | V rows of the text tokens of every branch (T <= 80 keys); rows [b*rows_per_branch, (b+1)*rows_per_branch) of X attend to branch b.
 *   stats_out (may be NULL): fp32 [M][C/160][2] row statistics of Y for a following folded LayerNorm.
 *   workspace: univst_attn2_fused_workspace_bytes(B, heads, head_dim) (K / V of every (branch, head) in operand order).
 * Served: C = 320 with 8 heads (the 64x64 level of SD-v1.x), rows_per_branch % 64 == 0; else UNIVST_ERR_ARG. */
int64_t univst_attn2_fused_workspace_bytes(int B, int heads, int head_dim);
int univst_attn2_fused(const void* X, int64_t ldx, const float* ln_stats, float ln_eps, const float* ln_wsum, const float* ln_bias,
                       const void* Wq_frag, int q_prescaled, const void* kv, int B, int T, int64_t rows_per_branch, const void* Wo_frag,
                       const void* bias_o, const void* residual, int64_t ldr, void* Y, int64_t ldy, int64_t M, int C, int heads,
                       float* stats_out, void* workspace, void* stream);
/* The same with the SELF-attention's out projection in front (attention.py:316-327: attn1.to_out + hidden_states -> norm2 -> attn2 -> + hidden_states):
 *     H2 = attn_out Wp^T + bias_p + residual_in;   Y = to_out( attention( LN(H2) Wq^T, text K, text V ) ) + bias_o + H2
 * H2 — which only this block reads — stays in LDS; its LayerNorm statistics are taken inside (ln_wsum / ln_bias as in univst_linear_ln, Wq_frag from the
 * gamma-folded weight).  Replaces univst_linear_ln (attn1.to_out, stats_out) + univst_attn2_fused.  Same shapes as univst_attn2_fused. */
int univst_attn12_fused(const void* attn_out, int64_t ldx, const void* Wp_frag, const void* bias_p, const void* residual_in, int64_t ldr_in, float ln_eps,
                        const float* ln_wsum, const float* ln_bias, const void* Wq_frag, int q_prescaled, const void* kv, int B, int T,
                        int64_t rows_per_branch, const void* Wo_frag, const void* bias_o, void* Y, int64_t ldy, int64_t M, int C, int heads,
                        float* stats_out, void* workspace, void* stream);
/* W [N][K] (N % 16 == 0, K % 32 == 0) -> the order in which v_mfma_f32_16x16x32_f16 takes it as its A operand: [N/16][K/32][64 lanes][8 halfs],
 * lane (l15, g) = W[nf*16 + l15][ks*32 + g*8 .. +8] — one contiguous 1 KB load per fragment for kernels that read weights straight into registers. */
int univst_frag_pack(const void* W, void* out, int N, int K, void* stream);
/* NHWC implicit-GEMM conv: taps 9 (3x3, pad 1) or 1; optional second source (channel concat), fused nearest x2
 * upsample of the input, stride 1/2.  W is [Cout][taps][C1+C2].  Replaces resnet.py:57-80,145,226. */
int univst_conv_nhwc(const void* X1, const void* X2, int C1, int C2, int imgs, int Hs, int Ws, int upsample, int stride,
                     int taps, const void* W, const void* bias, const void* rowbias, int rows_per_rowbias,
                     const void* residual, void* Y, int Cout, void* stream);
/* the same 3x3 conv with the "tap-inner" weight layout [Cout][(C1+C2)/64][9][64] (C1, C2 multiples of 64): the nine taps
 * of a 64-channel slab are consecutive k tiles, which keeps the re-read activation rows in the XCD's L2. */
int univst_conv_nhwc_tapinner(const void* X1, const void* X2, int C1, int C2, int imgs, int Hs, int Ws, int upsample, int stride,
                              const void* W, const void* bias, const void* rowbias, int rows_per_rowbias,
                              const void* residual, void* Y, int Cout, void* stream);
/* the 3x3 / stride-1 conv (optionally over the nearest x2 upsampled input) from an input patch kept in LDS
 * (conv_patch_kernel): weight layout [Cout][(C1+C2)/32][9][32]
 * (C1, C2 multiples of 32, Cout multiple of 320, image width a multiple of 16, whole image rows per 256/192-row tile, at
 * least 150 tiles — otherwise UNIVST_ERR_ARG: the UNet graph passes both weight copies and falls back by itself).
 * Replaces resnet.py:57-80 for the 64x64 .. 16x16 levels. */
int univst_conv3x3_patch(const void* X1, const void* X2, int C1, int C2, int imgs, int Hs, int Ws, int upsample, const void* W32,
                         const void* bias, const void* rowbias, int rows_per_rowbias, const void* residual, void* Y, int Cout,
                         void* stream);
/* A GroupNorm folded into the linear that consumes its output (the per-frame GroupNorm -> proj_in of a transformer block, attention.py:121-123): per
 * statistics unit s (rows_per_stat rows) the weight set W_sets[s][n][k] = fp16(W[n][k] * gamma_k * rstd_{s,g(k)}) and the fp32 bias
 * bias32[s][n] = bias[n] + sum_k W[n][k] * (beta_k - mean_{s,g(k)} * rstd * gamma_k); univst_linear_sets then runs on the RAW rows with the set of each
 * row range (rows_per_set = rows_per_stat: a multiple of 256 or 192) — GroupNorm(X) W^T + bias without the normalised copy of X.  stats_out as in
 * univst_linear_ln.  Direct 256x320 problems only (N % 320 == 0, >= 150 tiles); else UNIVST_ERR_ARG. */
int univst_groupnorm_fold_linear(const void* X, int C, int64_t rows, int rows_per_stat, int groups, float eps, const void* gamma, const void* beta,
                                 const void* W, const void* bias, int N, void* W_sets, float* bias32, void* workspace, void* stream);
int univst_linear_sets(const void* X, int64_t ldx, const void* W_sets, const float* bias32, int rows_per_set, const void* residual, int64_t ldr,
                       void* Y, int64_t ldy, int M, int N, int K, float* stats_out, void* stream);
/* GroupNorm(+SiLU) on NHWC rows; rows_per_stat = F*H*W (5-D, stats across frames: resnet.py:338,369) or H*W
 * (per frame: attention.py:121).  workspace: univst_groupnorm_workspace_bytes(). */
int64_t univst_groupnorm_workspace_bytes(int64_t rows, int rows_per_stat, int groups);
int univst_groupnorm_nhwc(const void* X1, const void* X2, int C1, int C2, int64_t rows, int rows_per_stat, int groups,
                          float eps, const void* gamma, const void* beta, int silu, void* Y, void* workspace,
                          void* stream);
int univst_layernorm(const void* X, void* Y, const void* gamma, const void* beta, int64_t rows, int C, float eps,
                     void* stream);
/* multi-source flash attention.  q rows (bf*Nq+i) at ldq, k/v rows (src*Nkv+j) at ldkv, src_idx int32 [BF][nsrc].
 * Optional (may be NULL): src_cnt int32 [BF] = number of leading sources actually used for that frame, src_logw fp32
 * [BF][nsrc] = log2 multiplicity of a source — a frame that occurs m times in the reference's concatenated key set
 * ([prev, cur, first] at f = 0, 1) is read ONCE with log2(m) added to its scores, which is the same softmax.
 * q_prescaled != 0: q already carries the factor log2(e)/sqrt(head_dim) (the UNet graph folds it into the to_q weights
 * for head_dim 40, so the fastest kernel can take scale and running max inside the QK^T MFMA); 0: plain q.
 * Replaces attention.py:384-420, pnp_utils.py:59-92 and diffusers AttnProcessor2_0. */
int univst_attention(const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, void* out, int64_t ldo,
                     const int32_t* src_idx, const int32_t* src_cnt, const float* src_logw, int nsrc, int BF, int Nq, int Nkv,
                     int heads, int head_dim, int q_prescaled, void* stream);
/* the same attention with the key set of a query split over TWO launches (round 6: a rank of the frame shard consumes the key frames it holds while the
 * halo frames of attention.py:384-413 / pnp_utils.py:59-84 are still on the wire; online softmax does not depend on the order of the keys).
 *   phase 1: state_out != NULL, state_in == NULL — rows of `out` normalised over this launch's sources, state_out[((bf*heads + h)*Nq + i)*2] = (m, l):
 *            the reference of the exponentials in log2 units and the denominator w.r.t. it; a frame with src_cnt[bf] == 0 gets zero rows and l = 0;
 *   phase 2: state_in = phase 1's state_out, `out` = phase 1's rows (read-modify-write): out <- softmax over the UNION of both launches' sources —
 *            exact up to the fp16 rounding of the phase-1 rows; frames with src_cnt[bf] == 0 keep their phase-1 rows.
 * src_cnt is required.  Served at every head_dim of univst_attention. */
int univst_attention_phase(const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, void* out, int64_t ldo,
                           const int32_t* src_idx, const int32_t* src_cnt, const float* src_logw, int nsrc, int BF, int Nq, int Nkv,
                           int heads, int head_dim, int q_prescaled, float* state_out, const float* state_in, void* stream);
/* ---- first vertical slice of the SD3 / SD3.5 rectified-flow path (the reference-owned pieces only) ----
 * The joint-attention processors of backbones/video_diffusion_sd3/pnp_utils.py: CrossFrameProcessor (:17-131; shift = 0) and
 * AttentionShiftProcessor (:143-271; shift = 1 applies the AdaIN shift with alpha 0.8 / gamma 2.0 and the given beta.  The window test
 * eta1*50 <= idx <= eta2*50 and beta = 0.9 -> 0.1 over the window are evaluated by the CALLER in double, as the reference's Python does
 * (pnp_utils.py:183-186) — in fp32, eta1 = 0.3 makes 0.3f*50 = 15.000001 and step 15 falls out of the window — under the documented
 * fixed reading thresh2 == eta2: the reference reads an attribute it never sets).  hidden [B, N, Cin] image tokens of
 * B = (branches x clip_length) frames, enc [B, Nt, Cin] text tokens or NULL; keys of frame f = image tokens of ['first', f-1, f]
 * of its clip (read by pointer) ++ its text tokens (one extra key segment); out_img [B, N, Cin], out_txt [B, Nt, Cin].
 * clip_length == 0: no cross-frame gather — keys of frame f are its own image tokens ++ its text tokens (diffusers' stock
 * JointAttnProcessor2_0, what the MM-DiT runs before register_spatial_attention_pnp / with no processor set).
 * Weights are diffusers' Attention parameters, fp16 device pointers, [out, in] row-major; NULL = absent (biases, the RMS norms of
 * SD3-medium, to_add_out when context_pre_only). */
typedef struct {
    const void *to_q, *to_q_bias, *to_k, *to_k_bias, *to_v, *to_v_bias;             /* [heads*head_dim, Cin] */
    const void *norm_q, *norm_k;                                                    /* [head_dim] RMSNorm weights (SD3.5) */
    const void *add_q, *add_q_bias, *add_k, *add_k_bias, *add_v, *add_v_bias;       /* added (text) projections */
    const void *norm_added_q, *norm_added_k;
    const void *to_out, *to_out_bias, *to_add_out, *to_add_out_bias;                /* [Cin, heads*head_dim] */
} univst_sd3_attn_weights;
/* optional: the MM-DiT block's gated residuals fused into the two out-projections (diffusers JointTransformerBlock:
 * hidden + gate_msa[:, None] * attn_output): out_img = res_img + gate_img[b] (.) to_out(o), out_txt likewise through to_add_out.
 * gate_*: rows of Cin halfs, ld_gate_* apart (chunks of the two adaLN linears' outputs), one per frame b. */
typedef struct {
    const void *res_img, *gate_img, *res_txt, *gate_txt;
    int64_t ld_gate_img, ld_gate_txt;
} univst_sd3_gated_residual;
/* the window test and beta of AttentionShiftProcessor in double (pnp_utils.py:183-186, fixed reading thresh2 == eta2), for callers outside Python:
 * active = eta1*50 <= idx <= eta2*50, beta = 0.9 -> 0.1 over the window (0 outside).  Pass (shift = active, beta) to univst_sd3_joint_attention. */
int univst_sd3_shift_window(int idx, double eta1, double eta2, int* active, float* beta);
int univst_sd3_joint_attention(const univst_sd3_attn_weights* w, const void* hidden, const void* enc, int B, int N, int Nt, int Cin,
                               int heads, int head_dim, int clip_length, int shift, float beta, float rms_eps,
                               void* out_img, void* out_txt, const univst_sd3_gated_residual* gated_residual /* may be NULL */,
                               univst_comm* comm /* may be NULL */, void* stream);
/* comm (frame shard, world > 1): the batch holds frames [rank*clip_length, (rank+1)*clip_length) of every branch; K | V of the clip's
 * first frame (from rank 0) and of the frame before this rank's first one (from rank - 1) arrive through the communicator's peer
 * writes (one exchange + one one-float all-reduce as a barrier per call; workspace >= 64 KiB + 6 x branches*N*Cin fp16: since round 6 a pack is the layer's HIDDEN rows of a boundary frame, posted on the communicator's forked
 * stream at the start of the op; the receiving rank projects them to K | V, applies the k RMSNorm and the shift, and attends in two phases). */
/* the shift alone, in place on a fused [3*F*N, 3C] q | k | v buffer (branch 0 content, 1 style, 2 stylised; row stride ld):
 * q2 <- gamma*(alpha*q0 + (1-alpha)*q2);  k2 <- beta*AdaIN(k2; k1) + (1-beta)*k1 (same for v) with the SD3 plugin's AdaIN
 * (pnp_utils.py:289-302: F.instance_norm over (N, head_dim) jointly per (frame, head), style mean / unbiased std per channel over N).
 * ws: 4*F*2C + 2*F*2*heads floats. */
int univst_sd3_adain_shift(void* qkv, int64_t ld, int F, int N, int C, int heads, float alpha, float beta, float gamma, void* ws,
                           void* stream);
/* diffusers RMSNorm over the head dim, in place: x[r, h*d + e] *= rsqrt(mean_e x^2 + eps) * weight[e]  (row stride ld) */
int univst_rmsnorm_heads(void* x, int64_t ld, int64_t rows, int heads, int head_dim, const void* weight, float eps, void* stream);
/* y = LayerNorm(x; no affine, eps) * (1 + scale[b]) + shift[b]: AdaLayerNormZero / ZeroX / Continuous of the MM-DiT blocks (diffusers
 * normalization.py, third-party).  x, y [rows, C] (C % 8 == 0, <= 4096); scale / shift are rows of a [rows / rows_per_batch, ...]
 * matrix ld_mod halfs apart (chunks of the adaLN linear's output).  y2 / scale2 / shift2 (all NULL or all set): a second modulation
 * of the same normalised rows (AdaLayerNormZeroX, the dual-attention blocks of SD3.5-medium). */
int univst_adaln_modulate(const void* x, void* y, const void* scale, const void* shift, int64_t ld_mod, int64_t rows, int64_t rows_per_batch,
                          int C, float eps, void* y2, const void* scale2, const void* shift2, void* stream);
/* out = x + gate[b] * y: the gated residuals of diffusers' JointTransformerBlock (gate rows ld_gate halfs apart; out may alias x) */
int univst_gate_residual(const void* x, const void* gate, int64_t ld_gate, const void* y, void* out, int64_t rows, int64_t rows_per_batch, int C,
                         void* stream);
/* elementwise activation, fp32 arithmetic (n % 8 == 0; out may alias x): SiLU (adaLN / timestep / pooled-text MLPs) and GELU(tanh)
 * (FeedForward activation_fn="gelu-approximate") */
#define UNIVST_ACT_SILU 0
#define UNIVST_ACT_GELU_TANH 1
#define UNIVST_ACT_NONE (-1)
int univst_activation(const void* x, void* out, int64_t n, int act, void* stream);
/* diffusers Timesteps / get_timestep_embedding: t fp32 [B] (device) -> out fp16 [B, dim] = [sin | cos](t * freq) (halves swapped when
 * flip_sin_to_cos), freq_i = exp(-ln(max_period) i / (dim/2 - downscale_freq_shift)), fp32 arithmetic */
int univst_timestep_embedding(const float* t, void* out, int B, int dim, int flip_sin_to_cos, float downscale_freq_shift, float max_period,
                              void* stream);
/* PatchEmbed's strided conv as a linear: latents [B, C, H, W] -> rows [B*(H/p)*(W/p), C*p*p] in the k order of the flattened conv
 * weight; and the inverse for proj_out's rows [.., p*p*C] ((u, v, c) order) -> latents (transformer_3D_model.py:95-103) */
int univst_sd3_patchify(const void* latents, void* rows, int B, int C, int H, int W, int patch, void* stream);
int univst_sd3_unpatchify(const void* rows, void* latents, int B, int C, int H, int W, int patch, void* stream);
/* out = a*x + b*y + c*z (fp16 storage, fp32 arithmetic): the updates of rf_inversion / rf_solver (flow_inversion.py:123-264) */
int univst_axpbypcz(const void* x, const void* y, const void* z, void* out, float a, float b, float c, int64_t n, void* stream);

/* AdaIN-guided attention shift in place on the fused QKV buffer [3*F*N, 3C]; stats_ws: 4*F*2C floats.
 * Replaces pnp_utils.py:47-57 + attention_adain :114-125. */
int univst_attention_adain_shift(void* qkv, int64_t ld, int F, int N, int C, float alpha, float beta, float gamma,
                                 void* stats_ws, void* stream);
/* latent_adain on [1,C,F,H,W] (pnp_utils.py:128-139) */
int univst_latent_adain(const void* cnt, const void* sty, void* out, int C, int F, int HW, void* stream);
/* the same in two halves for frame shards: stats[c] = {sum, sumsq} of the content over the LOCAL (F,H,W) (the caller
 * all-reduces them), then apply with the global element count n_total = F_total*H*W */
int univst_latent_adain_stats(const void* cnt, float* stats2c, int C, int F, int HW, void* stream);
int univst_latent_adain_apply(const void* cnt, const void* sty, const float* stats2c, int64_t n_total, void* out, int C,
                              int F, int HW, void* stream);
/* out = cx*x + ce*eps: DDIMScheduler.step / next_step with host-folded coefficients (stable_diffusion.py:761,
 * ddim_inversion.py:190-204) */
int univst_axpby(const void* x, const void* eps, void* out, float cx, float ce, int64_t n, void* stream);
/* out = (1-m)*a + m*b, m [F,h,w] broadcast over C (stable_diffusion.py:687-702); m NULL => m = 0 */
int univst_mask_blend(const void* a, const void* b, const void* m, void* out, int C, int64_t FHW, void* stream);
/* uint8 {0,1} mask [F,H,W] -> fp16 [F,h,w], torch bilinear align_corners=False (stable_diffusion.py:689) */
int univst_mask_resize(const uint8_t* mask, void* out, int F, int H, int W, int h, int w, void* stream);

/* ------------------------------------------------------------------ mask propagation (src/mask_propagation.py:72-83,60-69) */
/* one target frame: feat_tar [hw,C] f32, feat_src [Nsrc,C] f32 (row-major, un-normalised), segs_src [ncls,Nsrc] f32
 * -> segs_tar [ncls,hw] f32.  workspace: univst_maskprop_workspace_bytes(). */
int64_t univst_maskprop_workspace_bytes(int hw, int Nsrc, int C);
int univst_maskprop_frame(const float* feat_tar, const float* feat_src, const float* segs_src, float* segs_tar, int hw,
                          int Nsrc, int C, int ncls, float temperature, int topk, void* workspace, void* stream);
/* segs [ncls,h,w] f32 -> bilinear up to [H,W], per-class min-max, first-max argmax, !=0 -> 255 : uint8 [H,W] */
int univst_maskprop_finalize(const float* segs, uint8_t* mask_out, int ncls, int h, int w, int H, int W, void* workspace,
                             void* stream);

/* ------------------------------------------------------------------ flow warp + occlusion + window blend
 * (src/cal_optica_flow.py:20-46, stable_diffusion.py:731-751) */
/* acc[H,W,3] f32 += get_warp(key, now): occlusion ||(c+fwd)+bwd-c|| > thr, cv2.remap-exact bilinear warp of `now`
 * by fwd, occluded pixels take `key`.  key/now uint8 [H,W,3], fwd/bwd f32 [H,W,2]. */
int univst_warp_accumulate(const uint8_t* key, const uint8_t* now, const float* fwd, const float* bwd, float* acc, int H,
                           int W, float threshold, void* stream);
/* The whole window mean of ONE key frame in one launch (stable_diffusion.py:731-747, the inner loop over bias = -r .. r): frames
 * uint8 [F,H,W,3] is the Gauss-Seidel working copy, frame `key` is replaced in place by trunc(mean of {itself, get_warp(key, key+b)});
 * flows: HOST array of 2*nn device pointers, fp32 [H,W,2] each — (forward key -> key+b, backward key+b -> key) of the nn in-clip
 * neighbours in increasing b (r <= 4).  Same arithmetic as univst_warp_accumulate + univst_accumulate_u8 + univst_window_store
 * (bit-identical), 1 launch instead of 2r + 2. */
int univst_warp_window_key(uint8_t* frames, const float* const* flows, int nn, int F, int H, int W, int key, int r, float threshold,
                           void* stream);
/* latent-space sliding window (no reference code exists for it — definition in csrc/warp.hip and DESIGN.md):
 * x0 [C,F,h,w] fp16 in place, lflow [F, 2r+1, h, w, 2] fp32 = flow from frame k to frame k+b in latent-pixel units. */
int univst_latent_window_smooth(void* x0, const float* lflow, int C, int F, int h, int w, int r, float thr, void* stream);
int univst_accumulate_u8(const uint8_t* frame, float* acc, int64_t n, void* stream);
/* dst[i] = (uint8) trunc(acc[i] / weight) */
int univst_window_store(const float* acc, float weight, uint8_t* dst, int64_t n, void* stream);

/* ------------------------------------------------------------------ per-kernel-class HIP-event timing (bench.py roofline leg)
 * classes (one per kernel symbol): 0 gemm_big<0>, 1 gemm_big<1> (conv), 2 gemm_kernel<*,0> (linear), 3 gemm_kernel<*,1>
 * (conv), 4 attention d=40, 5 attention d=80, 6 other attention, 7 groupnorm, 8 layernorm, 9 adain shift, 10 text attention,
 * 11 conv_patch_kernel (LDS-patch 3x3 convs), 12 attn2_fused_kernel.
 * While enabled every launch of these classes is bracketed by hipEvents on its stream; collect() waits for
 * them and returns per class: summed ms, launch count, algorithmic flops and algorithmic bytes. */
#define UNIVST_PROFILE_CLASSES 13
int univst_profile_enable(int on);
int univst_profile_collect(double* ms, int64_t* count, double* flops, double* bytes, int nclasses);
/* per class, for the records the LAST univst_profile_collect returned: the EXPANDED operand bytes of the GEMM classes — for a 3x3 conv `bytes` above
 * charges input + weights + output (what has to cross HBM once), this figure charges the im2col operand M x 9Cin the matrix pipe consumes */
int univst_profile_collect_aux(double* aux_bytes, int nclasses);
/* ';'-joined kernel symbols launched in class `cls` since profiling was last switched on (template arguments without spaces, e.g.
 * "gemm_big_kernel<0,4,2>"; classes whose launchers do not name their kernels give ""): bench.py quotes PMC traffic only when every one of them
 * has an entry in the committed counter file */
int univst_profile_symbols(int cls, char* buf, int n);

/* bring-up aid: what ds_read_b64_tr_b16 returns per lane for a known LDS image (256 floats) */
int univst_debug_tr16(float* out256, void* stream);
/* measurement aid (bench.py --emulate-wire): a one-thread kernel that keeps `stream` busy for `us` microseconds — the stand-in for a transfer of
 * bytes / link rate when one rank of a multi-GPU job is emulated on a 1-GPU box */
int univst_debug_delay_us(double us, void* stream);

#ifdef __cplusplus
}
#endif
#endif
