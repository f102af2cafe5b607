// Frame-shard communicator inside the library (SURVEY §8b `univst_comm_*`, §8e): one process per GPU of ONE node; every rank owns a
// fine-grained device region that its peers map through HIP IPC (dmabuf handles), and the three couplings of the sharded UNet are
// device-side peer writes + flags over xGMI — no host callback, no Python, no second stream, nothing a hipGraph cannot hold:
//
//   all-reduce (45 x [3, 32, 2] fp32 per UNet call: the 5-D GroupNorm statistics; latent_adain's [4, 2]):  ONE one-block kernel per
//     call — every rank writes its vector into slot[parity][rank] of EVERY peer's region, raises flag[parity][rank] = epoch there,
//     then waits until all `world` flags of its own region carry the epoch and sums the slots in RANK ORDER (bitwise the same
//     result on every rank, independent of arrival order).  Latency = one xGMI write + one poll, not a ring of 2(W-1) hops.
//   K/V halo + first-frame broadcast (16 per UNet call, up to 15.7 MB packs):  a copy kernel writes this rank's last-frame pack into
//     rank+1's inbox and rank 0's first-frame pack into every peer's inbox (7 different links of the xGMI mesh at once), a
//     one-thread kernel then raises the flags; the receiver's one-block wait kernel precedes its unpack in stream order.  Not an
//     all-gather of all K/V: 8x fewer bytes on a per-link-bound fabric.
//
// Reuse without acknowledgements: epochs only grow, slots / inboxes are double-buffered by epoch parity, and a rank can never be two
// all-reduces ahead of a peer (it needs the peer's contribution to the one in between), nor two K/V exchanges (>= 2 GroupNorm
// all-reduces separate consecutive exchanges; the receiver's unpack precedes its next all-reduce in stream order).
// Memory model: payload and flags live in hipDeviceMallocFinegrained memory; payload words of the all-reduce are system-scope
// relaxed atomics, flags system-scope release / acquire; the K/V payload is written by a whole kernel and published by the NEXT
// kernel in stream order (kernel-boundary release), consumed by the kernel AFTER the wait kernel (kernel-boundary acquire).
// Every spin is bounded: a rank that gives up writes a code into a host-mapped status word, which the next API call reports
// (UNIVST_ERR_STATE) instead of hanging the node.
// Validated on a 1-GPU box by two PROCESSES sharing the GPU (tests/test_gpu_unet.py::*ipc*; tools/probes/ipc_probe.hip measured a
// 1.2 us flag round trip between kernels of two processes); the same primitives are what RCCL itself uses between GPUs of a node.
#include <string.h>

#include "common.h"
#include "kernels.h"
#include "unet.h"

int uv_launch_delay_us(double us, hipStream_t s);

namespace {

constexpr int UV_COMM_MAXW = 8;
constexpr int UV_AR_MAX = 1024;                              // floats per all-reduce
constexpr long UV_OFF_FLAGS = 0;                             // u32 ar_flag[2][8] | kv_flag[2][2] (prev, first) at word 32
constexpr long UV_OFF_AR = 4096;                             // float ar_slot[2][8][UV_AR_MAX]
constexpr long UV_OFF_WS = UV_OFF_AR + 2L * UV_COMM_MAXW * UV_AR_MAX * 4;      // the UNet's comm workspace (64 KiB + K/V slots)
constexpr long UV_SPIN_LIMIT = 40000000L;                    // polls (>= 0.5 us each) before giving up: tens of seconds

struct Peers {
    char* p[UV_COMM_MAXW];
};

__device__ __forceinline__ bool spin_until(const unsigned* flag, unsigned epoch) {
    long spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > UV_SPIN_LIMIT) return false;
    }
    return true;
}

// in-place SUM over ranks of n <= UV_AR_MAX floats at `buf`.  One block of 256 threads.
__global__ __launch_bounds__(256) void comm_allreduce_kernel(Peers peers, int rank, int world, unsigned epoch, float* buf, int n,
                                                             int* status) {
    const int tid = threadIdx.x, par = epoch & 1;
    for (int p = 0; p < world; ++p) {
        float* dst = reinterpret_cast<float*>(peers.p[p] + UV_OFF_AR) + ((long)par * UV_COMM_MAXW + rank) * UV_AR_MAX;
        for (int i = tid; i < n; i += 256) __hip_atomic_store(dst + i, buf[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
    __syncthreads();
    __shared__ int ok;
    if (tid == 0) ok = 1;
    __syncthreads();
    if (tid < world) {
        unsigned* f = reinterpret_cast<unsigned*>(peers.p[tid] + UV_OFF_FLAGS) + par * UV_COMM_MAXW + rank;
        __hip_atomic_store(f, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned* mine = reinterpret_cast<const unsigned*>(peers.p[rank] + UV_OFF_FLAGS) + par * UV_COMM_MAXW + tid;
        if (!spin_until(mine, epoch)) {
            ok = 0;
            *status = 100 + tid;                  // peer `tid` never arrived
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    if (!ok) return;
    const float* slots = reinterpret_cast<const float*>(peers.p[rank] + UV_OFF_AR) + (long)par * UV_COMM_MAXW * UV_AR_MAX;
    for (int i = tid; i < n; i += 256) {
        float s = 0.f;
        for (int r = 0; r < world; ++r) s += __hip_atomic_load(slots + (long)r * UV_AR_MAX + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        buf[i] = s;
    }
}

// 16-byte copies of `nvec` uint4 from src to up to 7 destinations (peer memory): the source is read once
struct Dsts {
    uint4* d[UV_COMM_MAXW];
    int n;
};
__global__ __launch_bounds__(256) void comm_multicast_kernel(const uint4* __restrict__ src, Dsts dsts, long nvec) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
        const uint4 v = src[i];
        for (int k = 0; k < dsts.n; ++k) dsts.d[k][i] = v;
    }
    // (round 4, belt and braces for the first contact with a real multi-GPU node: every thread drains its payload stores and writes them back at
    // SYSTEM scope before its wave ends — the publication no longer rests on the end-of-kernel release + the next kernel's fence alone; on one GPU
    // the two are indistinguishable, and the cost is one fence per thread on a kernel that moves megabytes)
    __threadfence_system();
}
struct Flags {
    unsigned* f[UV_COMM_MAXW + 1];
    int n;
};
__global__ void comm_raise_kernel(Flags fl, unsigned epoch) {
    __threadfence_system();
    if ((int)threadIdx.x < fl.n) __hip_atomic_store(fl.f[threadIdx.x], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void comm_wait_kernel(const unsigned* f0, const unsigned* f1, unsigned epoch, int* status) {
    const unsigned* f = threadIdx.x == 0 ? f0 : f1;
    if (f && !spin_until(f, epoch)) *status = 200 + threadIdx.x;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

}  // namespace

struct univst_comm {
    int rank = 0, world = 1;
    char* mine = nullptr;
    long bytes = 0, ws_bytes = 0;
    char* peer[UV_COMM_MAXW] = {};
    bool opened[UV_COMM_MAXW] = {};
    bool connected = false;
    unsigned ar_epoch = 0, kv_epoch = 0;
    int* status = nullptr;                                   // host-mapped: 0 ok, 100 + r / 200 + k = gave up waiting
    hipStream_t stream = nullptr;                            // the stream of the forward() in flight (callbacks carry none)
    // EMULATED rank (bench.py --emulate-rank r/w --comm-emulated; univst_comm_connect_emulated): rank r of a `world`-rank job ALONE on one GPU — every
    // peer pointer is this rank's own region, a post is a delay kernel of latency + (packs on this rank's busiest link) x bytes / rate followed by a raise
    // of this rank's OWN inbox flags, an all-reduce runs over this rank alone after one flag round trip of delay.  The kernels, streams and flag waits
    // are the production ones; the payload is whatever the inbox holds (zeros): timing only.
    bool emulated = false;
    double emu_gbps = 0.0, emu_lat_us = 0.0, emu_wire_us = 0.0;
    hipStream_t xstream = nullptr;                           // forked stream of callers without one of their own (the SD3 joint attention: uv_comm_fork)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
};

static int comm_check(univst_comm* c) {
    if (c->status && *c->status) {
        uv_set_error("frame-shard communicator: rank %d gave up waiting for a peer (code %d: 100 + r = all-reduce contribution of rank r, "
                     "200 / 201 = K/V halo / first-frame flag); a peer died or ran a different collective sequence", c->rank, *c->status);
        return UV_ERR_STATE;
    }
    return UV_OK;
}

int uv_comm_allreduce(univst_comm* c, float* buf, int n, hipStream_t s) {
    UV_REQUIRE(c && c->connected, "comm_allreduce: communicator not connected");
    UV_REQUIRE(buf && n >= 1 && n <= UV_AR_MAX, "comm_allreduce: n=%d outside 1..%d", n, UV_AR_MAX);
    int rc = comm_check(c);
    if (rc) return rc;
    Peers pe;
    for (int i = 0; i < UV_COMM_MAXW; ++i) pe.p[i] = c->peer[i];
    if (c->emulated) {              // one flag round trip, then the kernel over this rank alone (as rank 0 of a world of 1 on its own region)
        if (c->emu_lat_us > 0.0) {
            int drc = uv_launch_delay_us(c->emu_lat_us, s);
            if (drc) return drc;
            c->emu_wire_us += c->emu_lat_us;
        }
        hipLaunchKernelGGL(comm_allreduce_kernel, dim3(1), dim3(256), 0, s, pe, 0, 1, ++c->ar_epoch, buf, n, c->status);
        UV_LAUNCH_CHECK();
        return UV_OK;
    }
    hipLaunchKernelGGL(comm_allreduce_kernel, dim3(1), dim3(256), 0, s, pe, c->rank, c->world, ++c->ar_epoch, buf, n, c->status);
    UV_LAUNCH_CHECK();
    return UV_OK;
}

// callbacks with the signatures of univst_allreduce_fn / univst_kv_exchange_fn: the UNet graph drives the native communicator through
// the same two hooks a host-side (torch.distributed) communicator uses
static int comm_allreduce_cb(void* user, int64_t byte_off, int count) {
    univst_comm* c = (univst_comm*)user;
    return uv_comm_allreduce(c, reinterpret_cast<float*>(c->mine + UV_OFF_WS + byte_off), count, c->stream);
}
// Round 6: the exchange in two halves.  POST (pack already in the send slots): multicast to the peers + raise their flags, on `x` — the UNet graph's FORKED
// stream, so that the transfer runs beside the rank's own q|k|v projection and the local phase of its attention (unet.hip Fwd::kv_post).  WAIT: the
// one-block kernel that spins on this rank's own flags, on `s` — the forward's stream, in front of the first kernel that reads the inbox (Fwd::kv_join).
// The two streams meet only through memory (flags), never through an event per exchange: a cross-queue event costs ~15 us each way on this runtime, sixteen
// times per step.  Send-slot reuse needs no acknowledgement either: between two posts lie >= 2 GroupNorm all-reduces, and an all-reduce completes only after
// every peer contributed, which a peer does (stream order) only after it consumed this rank's previous pack.
static int comm_kv_post(univst_comm* c, int64_t o_send, int64_t o_first, int64_t o_prev, int64_t o_rfirst, int64_t nbytes, hipStream_t x) {
    int rc = comm_check(c);
    if (rc) return rc;
    UV_REQUIRE(nbytes % 16 == 0, "kv_exchange: pack size must be a multiple of 16 bytes");
    const unsigned epoch = ++c->kv_epoch;
    const int par = epoch & 1;
    auto flag = [&](int r, int which) { return reinterpret_cast<unsigned*>(c->peer[r] + UV_OFF_FLAGS) + 32 + par * 2 + which; };
    if (c->emulated) {
        // the slowest transfer this rank waits for: one pack per link (the first-frame pack from rank 0, the halo pack from rank - 1), except on rank 1
        // whose one link from rank 0 carries both; rank 0 only sends (its outgoing links carry one pack each)
        const double us = c->emu_lat_us + (c->rank == 1 ? 2.0 : 1.0) * (double)nbytes / (c->emu_gbps * 1e3);
        c->emu_wire_us += us;
        rc = uv_launch_delay_us(us, x);
        if (rc) return rc;
        Flags fe;
        fe.n = 2;
        fe.f[0] = flag(c->rank, 0);
        fe.f[1] = flag(c->rank, 1);
        hipLaunchKernelGGL(comm_raise_kernel, dim3(1), dim3(64), 0, x, fe, epoch);
        UV_LAUNCH_CHECK();
        return UV_OK;
    }
    const long nvec = nbytes / 16;
    const unsigned grid = (unsigned)((nvec + 256 * 8 - 1) / (256 * 8) < 1024 ? (nvec + 256 * 8 - 1) / (256 * 8) : 1024);
    Flags fl;
    fl.n = 0;
    if (c->rank < c->world - 1) {               // 1-hop halo: my last frame -> rank + 1
        Dsts d;
        d.n = 1;
        d.d[0] = reinterpret_cast<uint4*>(c->peer[c->rank + 1] + UV_OFF_WS + o_prev);
        hipLaunchKernelGGL(comm_multicast_kernel, dim3(grid), dim3(256), 0, x, reinterpret_cast<const uint4*>(c->mine + UV_OFF_WS + o_send), d, nvec);
        fl.f[fl.n++] = flag(c->rank + 1, 0);
    }
    if (c->rank == 0) {                         // the clip's first frame -> every other rank
        Dsts d;
        d.n = 0;
        for (int r = 1; r < c->world; ++r) {
            d.d[d.n++] = reinterpret_cast<uint4*>(c->peer[r] + UV_OFF_WS + o_rfirst);
            fl.f[fl.n++] = flag(r, 1);
        }
        hipLaunchKernelGGL(comm_multicast_kernel, dim3(grid), dim3(256), 0, x, reinterpret_cast<const uint4*>(c->mine + UV_OFF_WS + o_first), d, nvec);
    }
    if (fl.n) hipLaunchKernelGGL(comm_raise_kernel, dim3(1), dim3(64), 0, x, fl, epoch);
    UV_LAUNCH_CHECK();
    return UV_OK;
}
// The same exchange with its two packs posted SEPARATELY — different sizes, different moments (csrc/sd3.hip: the previous frame travels as hidden rows
// at the start of the layer, the clip's first frame as finished K | V once rank 0 has projected, normalised and shifted it): comm_kv_begin opens the
// exchange (one epoch for both packs), comm_kv_post_halo / comm_kv_post_first each multicast their pack and raise their own flag.  comm_kv_wait is common.
static int comm_kv_begin(univst_comm* c) {
    int rc = comm_check(c);
    if (rc) return rc;
    ++c->kv_epoch;
    return UV_OK;
}
static int comm_kv_post_part(univst_comm* c, int which, int64_t o_src, int64_t o_dst, int64_t nbytes, hipStream_t x) {       // which: 0 = halo (to rank + 1), 1 = first frame (rank 0 to all)
    UV_REQUIRE(nbytes % 16 == 0 && nbytes > 0, "kv_exchange: pack size must be a positive multiple of 16 bytes");
    const unsigned epoch = c->kv_epoch;
    const int par = epoch & 1;
    auto flag = [&](int r) { return reinterpret_cast<unsigned*>(c->peer[r] + UV_OFF_FLAGS) + 32 + par * 2 + which; };
    if (c->emulated) {             // the transfer INTO this rank, on the stream its own post runs on (packs on one link serialise there, as on rank 1's link)
        if (c->rank == 0) return UV_OK;
        const double us = c->emu_lat_us + (double)nbytes / (c->emu_gbps * 1e3);
        c->emu_wire_us += us;
        int rc = uv_launch_delay_us(us, x);
        if (rc) return rc;
        Flags fe;
        fe.n = 1;
        fe.f[0] = flag(c->rank);
        hipLaunchKernelGGL(comm_raise_kernel, dim3(1), dim3(64), 0, x, fe, epoch);
        UV_LAUNCH_CHECK();
        return UV_OK;
    }
    const long nvec = nbytes / 16;
    const unsigned grid = (unsigned)((nvec + 256 * 8 - 1) / (256 * 8) < 1024 ? (nvec + 256 * 8 - 1) / (256 * 8) : 1024);
    Dsts d;
    Flags fl;
    d.n = fl.n = 0;
    if (which == 0) {
        if (c->rank >= c->world - 1) return UV_OK;
        d.d[d.n++] = reinterpret_cast<uint4*>(c->peer[c->rank + 1] + UV_OFF_WS + o_dst);
        fl.f[fl.n++] = flag(c->rank + 1);
    } else {
        if (c->rank != 0) return UV_OK;
        for (int r = 1; r < c->world; ++r) {
            d.d[d.n++] = reinterpret_cast<uint4*>(c->peer[r] + UV_OFF_WS + o_dst);
            fl.f[fl.n++] = flag(r);
        }
    }
    hipLaunchKernelGGL(comm_multicast_kernel, dim3(grid), dim3(256), 0, x, reinterpret_cast<const uint4*>(c->mine + UV_OFF_WS + o_src), d, nvec);
    hipLaunchKernelGGL(comm_raise_kernel, dim3(1), dim3(64), 0, x, fl, epoch);
    UV_LAUNCH_CHECK();
    return UV_OK;
}
static int comm_kv_wait(univst_comm* c, hipStream_t s) {          // for the exchange the last comm_kv_post / comm_kv_begin opened
    if (c->rank == 0) return UV_OK;
    const unsigned epoch = c->kv_epoch;
    const int par = epoch & 1;
    unsigned* f = reinterpret_cast<unsigned*>(c->peer[c->rank] + UV_OFF_FLAGS) + 32 + par * 2;
    hipLaunchKernelGGL(comm_wait_kernel, dim3(1), dim3(2), 0, s, f, f + 1, epoch, c->status);
    UV_LAUNCH_CHECK();
    return UV_OK;
}
static int comm_kv_issue(univst_comm* c, int64_t o_send, int64_t o_first, int64_t o_prev, int64_t o_rfirst, int64_t nbytes, hipStream_t s) {
    int rc = comm_kv_post(c, o_send, o_first, o_prev, o_rfirst, nbytes, s);
    return rc ? rc : comm_kv_wait(c, s);
}

static int comm_kv_cb(void* user, int64_t o_send, int64_t o_first, int64_t o_prev, int64_t o_rfirst, int64_t nbytes) {
    univst_comm* c = (univst_comm*)user;
    return comm_kv_issue(c, o_send, o_first, o_prev, o_rfirst, nbytes, c->stream);
}

extern "C" {

int univst_comm_create(int rank, int world, int64_t ws_bytes, univst_comm** out) {
    UV_REQUIRE(out && world >= 1 && world <= UV_COMM_MAXW && rank >= 0 && rank < world, "comm_create: rank %d / world %d (max %d ranks: one node)", rank, world, UV_COMM_MAXW);
    UV_REQUIRE(ws_bytes >= (1 << 17), "comm_create: workspace of %lld bytes is too small (>= 128 KiB)", (long long)ws_bytes);
    univst_comm* c = new univst_comm();
    c->rank = rank;
    c->world = world;
    c->ws_bytes = ws_bytes;
    c->bytes = UV_OFF_WS + ((ws_bytes + 4095) & ~4095L);
    hipError_t e = hipExtMallocWithFlags((void**)&c->mine, (size_t)c->bytes, hipDeviceMallocFinegrained);
    if (e == hipSuccess) e = hipMemset(c->mine, 0, (size_t)c->bytes);
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->status, sizeof(int), hipHostMallocMapped);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        uv_set_error("comm_create: %s", hipGetErrorString(e));
        if (c->mine) (void)hipFree(c->mine);
        delete c;
        return UV_ERR_HIP;
    }
    *c->status = 0;
    c->peer[rank] = c->mine;
    if (world == 1) c->connected = true;
    *out = c;
    return UV_OK;
}

int univst_comm_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }

int univst_comm_export(univst_comm* c, void* handle_out) {
    UV_REQUIRE(c && handle_out, "comm_export: null argument");
    hipIpcMemHandle_t h;
    UV_HIP(hipIpcGetMemHandle(&h, c->mine));
    memcpy(handle_out, &h, sizeof(h));
    return UV_OK;
}

// handles: world x univst_comm_handle_bytes() bytes, rank-major (this rank's own entry is ignored).  Call on every rank after an
// out-of-band all-gather of the exported handles (torch.distributed on the Python side; any transport will do).
int univst_comm_connect(univst_comm* c, const void* handles) {
    UV_REQUIRE(c && handles, "comm_connect: null argument");
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank || c->opened[r]) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, (const char*)handles + (size_t)r * sizeof(h), sizeof(h));
        UV_HIP(hipIpcOpenMemHandle((void**)&c->peer[r], h, hipIpcMemLazyEnablePeerAccess));
        c->opened[r] = true;
    }
    c->connected = true;
    return UV_OK;
}

// same-process peers (host threads playing ranks on one GPU, tests): the regions are plain pointers
int univst_comm_connect_local(univst_comm* c, univst_comm* const* all) {
    UV_REQUIRE(c && all, "comm_connect_local: null argument");
    for (int r = 0; r < c->world; ++r) {
        UV_REQUIRE(all[r] && all[r]->world == c->world && all[r]->rank == r, "comm_connect_local: entry %d is not rank %d of this group", r, r);
        c->peer[r] = all[r]->mine;
    }
    c->connected = true;
    return UV_OK;
}

// bench.py --comm-emulated: this rank alone stands for rank `c->rank` of `c->world` (see struct univst_comm::emulated)
int univst_comm_connect_emulated(univst_comm* c, double link_gbps, double latency_us) {
    UV_REQUIRE(c, "comm_connect_emulated: null argument");
    UV_REQUIRE(link_gbps > 0.0 && latency_us >= 0.0 && latency_us < 1e5, "comm_connect_emulated: link rate %f GB/s, latency %f us", link_gbps, latency_us);
    for (int r = 0; r < c->world; ++r) c->peer[r] = c->mine;
    c->emulated = true;
    c->emu_gbps = link_gbps;
    c->emu_lat_us = latency_us;
    c->connected = true;
    return UV_OK;
}
int univst_comm_query(univst_comm* c, const char* name, double* out) {
    UV_REQUIRE(c && name && out, "comm_query: null argument");
    if (!strcmp(name, "emu_wire_us")) {          // modelled wire time issued since the last query (reading resets it)
        *out = c->emu_wire_us;
        c->emu_wire_us = 0.0;
        return UV_OK;
    }
    uv_set_error("comm_query: unknown quantity '%s'", name);
    return UV_ERR_ARG;
}

int univst_comm_destroy(univst_comm* c) {
    if (!c) return UV_OK;
    (void)hipDeviceSynchronize();
    for (int r = 0; r < c->world; ++r)
        if (c->opened[r]) (void)hipIpcCloseMemHandle(c->peer[r]);
    if (c->mine) (void)hipFree(c->mine);
    if (c->status) (void)hipHostFree(c->status);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->xstream) (void)hipStreamDestroy(c->xstream);
    delete c;
    return UV_OK;
}

int univst_comm_allreduce_f32(univst_comm* c, void* buf, int n, void* stream) {
    return uv_comm_allreduce(c, (float*)buf, n, (hipStream_t)stream);
}

int univst_comm_status(univst_comm* c) { return c && c->status ? *c->status : 0; }

}  // extern "C"

// ---- UNet side: route the graph's two hooks into the communicator (called by univst_unet_set_comm_native in abi.hip)
int uv_unet_attach_comm(UNet& u, univst_comm* c) {
    UV_REQUIRE(c && c->connected, "set_comm_native: communicator not connected");
    u.rank = c->rank;
    u.world = c->world;
    u.comm_ws = c->mine + UV_OFF_WS;
    u.comm_ws_bytes = c->ws_bytes;
    u.allreduce = comm_allreduce_cb;
    u.kv_exchange = comm_kv_cb;
    u.comm_user = c;
    u.native_comm = c;
    return UV_OK;
}
// ---- SD3 side (csrc/sd3.hip): the same K/V exchange with explicit offsets, plus a one-float all-reduce used as a barrier — the
// MM-DiT has no GroupNorm all-reduces between two consecutive exchanges, and it is those that keep a fast rank from overwriting an
// inbox its neighbour has not unpacked yet (see the reuse argument at the top of this file)
int uv_comm_kv_exchange(univst_comm* c, long o_send, long o_first, long o_prev, long o_rfirst, long nbytes, hipStream_t s) {
    UV_REQUIRE(c && c->connected, "kv_exchange: communicator not connected");
    UV_REQUIRE(o_send >= 0 && o_first >= 0 && o_prev >= 0 && o_rfirst >= 0 && nbytes > 0 &&
               (o_send > o_first ? o_send : o_first) + nbytes <= c->ws_bytes && (o_prev > o_rfirst ? o_prev : o_rfirst) + nbytes <= c->ws_bytes,
               "kv_exchange: a %ld-byte pack does not fit the communicator's %ld-byte workspace", nbytes, c->ws_bytes);
    c->stream = s;
    return comm_kv_cb(c, o_send, o_first, o_prev, o_rfirst, nbytes);
}
// the UNet graph's exchange in its two halves (see comm_kv_post)
int uv_comm_kv_post(univst_comm* c, long o_send, long o_first, long o_prev, long o_rfirst, long nbytes, hipStream_t x) {
    UV_REQUIRE(c && c->connected, "kv_exchange: communicator not connected");
    return comm_kv_post(c, o_send, o_first, o_prev, o_rfirst, nbytes, x);
}
int uv_comm_kv_begin(univst_comm* c) {
    UV_REQUIRE(c && c->connected, "kv_exchange: communicator not connected");
    return comm_kv_begin(c);
}
int uv_comm_kv_post_halo(univst_comm* c, long o_send, long o_prev, long nbytes, hipStream_t x) {
    UV_REQUIRE(c && c->connected, "kv_exchange: communicator not connected");
    return comm_kv_post_part(c, 0, o_send, o_prev, nbytes, x);
}
int uv_comm_kv_post_first(univst_comm* c, long o_first, long o_rfirst, long nbytes, hipStream_t x) {
    UV_REQUIRE(c && c->connected, "kv_exchange: communicator not connected");
    return comm_kv_post_part(c, 1, o_first, o_rfirst, nbytes, x);
}
int uv_comm_kv_wait(univst_comm* c, hipStream_t s) {
    UV_REQUIRE(c && c->connected, "kv_exchange: communicator not connected");
    return comm_kv_wait(c, s);
}
// a forked stream for callers that have none (csrc/sd3.hip): *x runs after everything queued on s so far; uv_comm_join makes s wait for what x holds (unused since the layer's barrier retires the fork: a cross-queue event into s cost 30x in the 2-process tests; kept for callers without a barrier)
int uv_comm_fork(univst_comm* c, hipStream_t s, hipStream_t* x) {
    static const int ov = getenv("UNIVST_KV_OVERLAP") ? atoi(getenv("UNIVST_KV_OVERLAP")) : 1;
    if (!ov) {
        *x = s;
        return UV_OK;
    }
    if (!c->xstream) {
        UV_HIP(hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking));
        UV_HIP(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        UV_HIP(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    }
    UV_HIP(hipEventRecord(c->ev_fork, s));
    UV_HIP(hipStreamWaitEvent(c->xstream, c->ev_fork, 0));
    *x = c->xstream;
    return UV_OK;
}
int uv_comm_join(univst_comm* c, hipStream_t s) {
    UV_HIP(hipEventRecord(c->ev_join, c->xstream));
    UV_HIP(hipStreamWaitEvent(s, c->ev_join, 0));
    return UV_OK;
}
// bench.py --emulate-wire: the same flag mechanics against a word of this process (unet.hip: a delay kernel + this raise on the forked stream stand in
// for a peer's multicast + raise; the forward's stream spins on the word exactly as it does on a peer-raised flag)
int uv_comm_launch_raise(unsigned* flag, unsigned epoch, hipStream_t s) {
    Flags fl;
    fl.n = 1;
    fl.f[0] = flag;
    hipLaunchKernelGGL(comm_raise_kernel, dim3(1), dim3(64), 0, s, fl, epoch);
    UV_LAUNCH_CHECK();
    return UV_OK;
}
int uv_comm_launch_wait(const unsigned* flag, unsigned epoch, int* status, hipStream_t s) {
    hipLaunchKernelGGL(comm_wait_kernel, dim3(1), dim3(2), 0, s, flag, (const unsigned*)nullptr, epoch, status);
    UV_LAUNCH_CHECK();
    return UV_OK;
}
int uv_comm_barrier(univst_comm* c, hipStream_t s) {             // (the first 64 KiB of the workspace are the all-reduce scratch of both paths)
    return uv_comm_allreduce(c, reinterpret_cast<float*>(c->mine + UV_OFF_WS), 1, s);
}
char* uv_comm_ws(univst_comm* c) { return c->mine + UV_OFF_WS; }
long uv_comm_ws_bytes(const univst_comm* c) { return c->ws_bytes; }
int uv_comm_rank(const univst_comm* c) { return c->rank; }
bool uv_comm_emulated(const univst_comm* c) { return c->emulated; }
int uv_comm_world(const univst_comm* c) { return c->world; }
void uv_comm_bind_stream(univst_comm* c, hipStream_t s) { c->stream = s; }
unsigned uv_comm_kv_parity(const univst_comm* c) { return (c->kv_epoch + 1) & 1; }
int uv_comm_poll(univst_comm* c) { return comm_check(c); }


// bench.py --emulate-wire: a kernel that occupies its stream for `us` microseconds (constant 100 MHz clock), standing in for a transfer of
// bytes / rate on a link this 1-GPU box does not have
namespace {
__global__ void delay_kernel(long ticks) {
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
}  // namespace
int uv_launch_delay_us(double us, hipStream_t s) {
    UV_REQUIRE(us >= 0.0 && us < 5e6, "delay: %f us", us);
    hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(1), 0, s, (long)(us * 100.0));
    UV_LAUNCH_CHECK();
    return UV_OK;
}
