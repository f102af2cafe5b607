// UNet handle internals (see unet.hip).
#pragma once
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/univst.h"
#include "common.h"

typedef univst_pnp univst_pnp_t;
struct UNet;
// comm.hip
int uv_unet_attach_comm(UNet& u, univst_comm* c);
void uv_comm_bind_stream(univst_comm* c, hipStream_t s);
unsigned uv_comm_kv_parity(const univst_comm* c);
int uv_comm_poll(univst_comm* c);
int uv_comm_allreduce(univst_comm* c, float* buf, int n, hipStream_t s);
int uv_comm_kv_exchange(univst_comm* c, long o_send, long o_first, long o_prev, long o_rfirst, long nbytes, hipStream_t s);
int uv_comm_kv_post(univst_comm* c, long o_send, long o_first, long o_prev, long o_rfirst, long nbytes, hipStream_t x);
int uv_comm_kv_begin(univst_comm* c);
int uv_comm_kv_post_halo(univst_comm* c, long o_send, long o_prev, long nbytes, hipStream_t x);
int uv_comm_kv_post_first(univst_comm* c, long o_first, long o_rfirst, long nbytes, hipStream_t x);
int uv_comm_kv_wait(univst_comm* c, hipStream_t s);
int uv_comm_fork(univst_comm* c, hipStream_t s, hipStream_t* x);
int uv_comm_join(univst_comm* c, hipStream_t s);
int uv_comm_launch_raise(unsigned* flag, unsigned epoch, hipStream_t s);
int uv_comm_launch_wait(const unsigned* flag, unsigned epoch, int* status, hipStream_t s);
int uv_comm_barrier(univst_comm* c, hipStream_t s);
char* uv_comm_ws(univst_comm* c);
long uv_comm_ws_bytes(const univst_comm* c);
int uv_comm_rank(const univst_comm* c);
bool uv_comm_emulated(const univst_comm* c);
int uv_comm_world(const univst_comm* c);

struct WTensor {
    half_t* ptr = nullptr;
    std::vector<long> shape;
};

struct Act {   // NHWC activation: [imgs, H, W, C] fp16
    half_t* p = nullptr;
    int imgs = 0, H = 0, W = 0, C = 0;
    const float* gst = nullptr;      // GroupNorm (sum, sumsq) per 16-row fragment and 10-channel sub-group, left by the producing conv / linear's epilogue ([C/10][rows/16][2]) or null
    long rows() const { return (long)imgs * H * W; }
};

struct Arena {   // first-fit allocator over one device slab; stream-ordered reuse
    struct Block {
        size_t off, size;
        bool free;
    };
    char* base = nullptr;
    size_t size = 0, high_water = 0;
    std::vector<Block> blocks;
    void* alloc(size_t bytes);
    void release(void* p);
    void reset();
};

struct UNet {
    univst_unet_cfg cfg;
    std::unordered_map<std::string, WTensor> weights, derived;
    std::unordered_map<long, int*> idx_tables;
    std::unordered_map<std::string, long> temb_off;    // resnet prefix -> column offset in the fused time_emb_proj (finalize)
    long temb_total = 0;
    Arena arena;
    bool finalized = false;
    int gn_producer = 1;      // GroupNorm statistics from the producing conv / linear's epilogue where the tile geometry allows (UNIVST_GN_PRODUCER=0: always the stand-alone pass)
    int chain_bands = 1;      // post-attention chain of a transformer block in row bands: 1 off (default: measured +0.4 ms per step in the graph), 0 auto (bands of >= 65536 rows), n > 1 forced (UNIVST_CHAIN_BANDS)
    int gn_fold = 1;          // the transformer blocks' per-frame GroupNorm folded into proj_in where the weight copies are cheap (round 5; 0: the apply pass)
    int attn2_fused = 2;      // the text cross-attention of a block as one launch where fused.hip serves the shape (round 5): 2 = with the self-attention's out projection in front, 1 = attn2 alone, 0 = q projection + attention + out projection
    int ln_fold = 2;          // transformer-block LayerNorms folded into the neighbouring linears: 0 none, 1 norm1 + norm2, 2 also norm3 (default since round 4; UNIVST_LN_FOLD)
    unsigned* d_counter = nullptr;
    // TRAINED temporal layers (fine-tuned 3-D checkpoints): the units whose *_temporal* parameters differ from the identity
    // initialisation (found on the device at finalize).  Empty for 2-D-initialised weights: the graph then skips them exactly.
    std::unordered_map<std::string, int> temporal_conv_active;     // conv prefix ("...resnets.0.conv1", "conv_in", ...)
    std::unordered_map<std::string, int> temporal_attn_active;     // transformer block prefix ("...transformer_blocks.0")
    std::string missing;
    // multi-GPU frame sharding hooks
    int rank = 0, world = 1;
    univst_allreduce_fn allreduce = nullptr;
    univst_kv_exchange_fn kv_exchange = nullptr;
    void* comm_user = nullptr;
    char* comm_ws = nullptr;
    long comm_ws_bytes = 0;
    struct univst_comm* native_comm = nullptr;   // set when the hooks above are the library's own IPC communicator (comm.hip)
    // round 6: the K/V exchange of a transformer block runs on a FORKED stream beside the block's own q|k|v projection and the local phase of its
    // attention (Fwd::kv_post / kv_join); kv_overlap = 0 issues it on the forward's stream instead (A/B aid).  emu_wire_*: bench.py --emulate-wire —
    // with a communicator that moves nothing (NullComm) a delay kernel on the forked stream stands in for the slowest transfer of each exchange.
    int kv_overlap = 1;
    int emu_wire_gbps = 0, emu_wire_lat_us = 3;
    double emu_wire_us = 0.0;                    // modelled wire time issued since the last univst_unet_query("emu_wire_us")
    hipStream_t xstream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool x_dirty = false;                        // the forked stream carries work of this forward that the forward's stream has not joined yet
    unsigned* emu_flag = nullptr;                // [0] the emulated inbox flag (device), [1] a status word for the wait kernel
    unsigned emu_epoch = 0;
    int comm_streams();                          // creates the forked stream + events on first use

    ~UNet();
    int load_tensor(const char* key, const void* dev_ptr, int dtype, const int64_t* shape, int ndim, hipStream_t s);
    int finalize(hipStream_t s);
    int reserve(int B, int F, int H, int W);
    int forward(const half_t* sample, float timestep, const half_t* text, int B, int F, int H, int W, int text_len,
                const univst_pnp_t* pnp, half_t* eps_out, half_t* feat_out, int ft_index, hipStream_t s);
    const WTensor* find(const std::string& k) const;
    half_t* W(const std::string& k);
    int derive_alloc(const std::string& k, std::vector<long> shape, half_t** out);
    int missing_error();
};
