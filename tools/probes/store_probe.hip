// Micro-probe (not part of the product): what bounds the store tail of a 256-row output tile?
//
// The GEGLU projection's epilogue stores 80 KB per CU and tile (256 rows x 160 fp16) and an in-kernel ablation prices that tail at
// ~6.4 us.  This probe replays the stores alone, with the kernel's lane -> address mapping and with
// alternatives, on ONE CU and on all 256 at once, with and without a matrix-pipe phase between tiles, and it times a global load
// issued right behind the stores (is the next tile's operand prefetch stuck behind them?).
//
//   pattern 0  wave (wm, wn) of 4 x 2 stores rows 64 wm .. +64, columns wn TW/2 .. +TW/2, 16 bytes per lane (the shipped slab epilogue)
//   pattern 1  wave w stores rows 32 w .. +32 over the full tile width (row segments of TW * 2 bytes)
//   pattern 2  the tile as one contiguous run (upper bound for any mapping)
//   pattern 3  pattern 0 with 8-byte stores (the round-3 register epilogue)
//   +4         the same with non-temporal stores
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

struct Args {
    _Float16* out; const float* src; long long* clk;
    int n_out;        // row stride of the output in halfs
    int tw;           // tile width in halfs (160 or 320)
    int tiles_n;      // column tiles per row of tiles
    int tiles;        // total tiles
    int mfma_iters;   // matrix-pipe phase between tiles (x 40 MFMAs per wave)
    int do_store;     // 0: no stores (phase only)
    int do_load;      // 1: one dependent global load per lane right behind the stores, its latency accumulated into clk[]
};

template <int PAT>
__global__ __launch_bounds__(512, 1) void store_probe(Args a) {
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    h8 av[10], bv[4];
    for (int i = 0; i < 10; ++i) for (int e = 0; e < 8; ++e) av[i][e] = (_Float16)(0.001f * ((tid + i + e) & 15));
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 8; ++e) bv[j][e] = (_Float16)(0.002f * ((tid + j + e) & 7));
    f4 acc[10][4];
    for (int i = 0; i < 10; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f4{0, 0, 0, 0};
    long long waited = 0; float sink = 0.f;
    constexpr bool NT = (PAT & 4) != 0;
    constexpr int P = PAT & 3;
    for (int t = blockIdx.x; t < a.tiles; t += gridDim.x) {
        for (int it = 0; it < a.mfma_iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 10; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        const int tm = t / a.tiles_n, tn = t % a.tiles_n;
        _Float16* base = a.out + (size_t)tm * 256 * a.n_out + (size_t)tn * a.tw;
        long long t0 = 0;
        if (a.do_load) t0 = __builtin_readcyclecounter();
        if (a.do_store) {
            if (P == 3) {
                const int cpr = a.tw / 2 / 4;                       // 8-byte chunks per row segment of this wave
                const int n = 64 * cpr / 64;
                for (int i = 0; i < n; ++i) {
                    const int e = i * 64 + lane, r = e / cpr, c = e % cpr;
                    h4 v; for (int q = 0; q < 4; ++q) v[q] = (_Float16)acc[i % 10][(i / 10) & 3][q];
                    h4* p = reinterpret_cast<h4*>(base + (size_t)((w >> 1) * 64 + r) * a.n_out + (w & 1) * (a.tw / 2) + c * 4);
                    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
                }
            } else {
                const int cpr = (P == 0 ? a.tw / 2 : a.tw) / 8;     // 16-byte chunks per row segment
                const int rows = P == 0 ? 64 : 32;
                const int n = rows * cpr / 64;
                for (int i = 0; i < n; ++i) {
                    const int e = i * 64 + lane, r = e / cpr, c = e % cpr;
                    h8 v; for (int q = 0; q < 8; ++q) v[q] = (_Float16)acc[i % 10][(i / 10) & 3][q & 3];
                    h8* p;
                    if (P == 0) p = reinterpret_cast<h8*>(base + (size_t)((w >> 1) * 64 + r) * a.n_out + (w & 1) * (a.tw / 2) + c * 8);
                    else if (P == 1) p = reinterpret_cast<h8*>(base + (size_t)(w * 32 + r) * a.n_out + c * 8);
                    else p = reinterpret_cast<h8*>(a.out + (size_t)t * 256 * a.tw + (size_t)(w * 32 * cpr + e) * 8);
                    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
                }
            }
        }
        if (a.do_load) {
            float x = __builtin_nontemporal_load(a.src + ((size_t)t * 512 + tid) * 16 % (1 << 24));
            sink += x;                                               // the add waits for the load
            asm volatile("" :: "v"(sink));
            waited += __builtin_readcyclecounter() - t0;
        }
    }
    float s = sink;
    for (int i = 0; i < 10; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0];
    if (s == 12345.678f) a.out[0] = (_Float16)s;
    if (a.do_load && lane == 0) a.clk[blockIdx.x * 8 + w] = waited;
}

static void launch(int pat, int grid, const Args& a) {
    switch (pat) {
        case 0: hipLaunchKernelGGL(store_probe<0>, dim3(grid), dim3(512), 0, 0, a); break;
        case 1: hipLaunchKernelGGL(store_probe<1>, dim3(grid), dim3(512), 0, 0, a); break;
        case 2: hipLaunchKernelGGL(store_probe<2>, dim3(grid), dim3(512), 0, 0, a); break;
        case 3: hipLaunchKernelGGL(store_probe<3>, dim3(grid), dim3(512), 0, 0, a); break;
        case 4: hipLaunchKernelGGL(store_probe<4>, dim3(grid), dim3(512), 0, 0, a); break;
        case 5: hipLaunchKernelGGL(store_probe<5>, dim3(grid), dim3(512), 0, 0, a); break;
        case 6: hipLaunchKernelGGL(store_probe<6>, dim3(grid), dim3(512), 0, 0, a); break;
        default: hipLaunchKernelGGL(store_probe<7>, dim3(grid), dim3(512), 0, 0, a); break;
    }
}

static float run(int pat, int grid, Args a, int reps = 3) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0); launch(pat, grid, a); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return best;
}

int main() {
    const int M = 196608;
    size_t bytes = (size_t)M * 2560 * 2;
    _Float16* out; float* src; long long* clk;
    hipMalloc(&out, bytes); hipMalloc(&src, (size_t)(1 << 24) * 4 + 64); hipMalloc(&clk, 256 * 8 * 8);
    hipMemset(out, 0, bytes); hipMemset(src, 0, (size_t)(1 << 24) * 4 + 64);
    const char* names[8] = {"wave quadrant 16 B", "wave row band 16 B", "tile contiguous 16 B", "wave quadrant 8 B",
                            "wave quadrant 16 B nt", "wave row band 16 B nt", "tile contiguous 16 B nt", "wave quadrant 8 B nt"};
    for (int tw = 160; tw <= 320; tw += 160) {
        const int n_out = tw == 160 ? 1280 : 1280;        // GEGLU output of the 64x64 level / a 1280-wide plain output
        const int tiles_n = n_out / tw, tiles_m = (tw == 160 ? M : M / 2) / 256, tiles = tiles_m * tiles_n;
        const double kb = 256.0 * tw * 2 / 1024;
        printf("\n=== tile 256 x %d halfs = %.0f KB, output row stride %d halfs, %d tiles (%.0f MB)\n", tw, kb, n_out, tiles, tiles * kb / 1024);
        // matrix phase calibrated to about 10 us per tile (the K = 320 main loop): 5 k tiles x 80 MFMAs per wave = 10 iterations
        for (int phase = 0; phase <= 10; phase += 10) {
            Args a{out, src, clk, n_out, tw, tiles_n, tiles, phase, 0, 0};
            float base1 = phase ? run(0, 1, Args{out, src, clk, n_out, tw, tiles_n, 200, phase, 0, 0}) : 0.f;
            float base256 = phase ? run(0, 256, a) : 0.f;
            printf("-- matrix phase per tile: %d x 40 MFMAs per wave; phase alone: one CU %.2f us per tile, 256 CUs %.2f us per tile\n", phase,
                   base1 * 1e3 / 200, base256 * 1e3 / (tiles / 256.0));
            for (int pat = 0; pat < 8; ++pat) {
                Args one{out, src, clk, n_out, tw, tiles_n, 200, phase, 1, 0};
                float t1 = run(pat, 1, one);
                a.do_store = 1;
                float tall = run(pat, 256, a);
                printf("   %-26s one CU %6.2f us per tile (%5.1f B/clk at 2.4 GHz)   256 CUs %6.2f us per tile  (%.2f TB/s)\n", names[pat],
                       t1 * 1e3 / 200, kb * 1024 / (t1 * 1e3 / 200 * 2400), tall * 1e3 / (tiles / 256.0), tiles * kb * 1024 / (tall * 1e-3) / 1e12);
            }
        }
        // a load issued right behind the stores
        for (int st = 0; st <= 1; ++st) {
            Args a{out, src, clk, n_out, tw, tiles_n, tiles, 10, st, 1};
            hipMemset(clk, 0, 256 * 8 * 8);
            float ms = run(0, 256, a, 1);
            long long h[256 * 8]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
            double sum = 0; for (int i = 0; i < 256 * 8; ++i) sum += (double)h[i];
            printf("   load behind %-10s: %.2f us per tile, mean store-issue + load wait %.0f cycles of the shader clock counter per tile and wave\n",
                   st ? "the stores" : "nothing", ms * 1e3 / (tiles / 256.0), sum / (256 * 8) / (tiles / 256.0));
        }
    }
    return 0;
}
