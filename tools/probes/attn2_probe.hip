// Micro-probe (not part of the product): where a block of attn2_fused_kernel (univst_amd/csrc/fused.hip) spends its time.  The kernel is
// compiled here with -DUV_A2_TRACE: lane 0 of every wave stores the cycle counter at the phase boundaries.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iunivst_amd/csrc -o tools/probes/attn2_probe tools/probes/attn2_probe.hip
#define UV_A2_TRACE 1
#include <stdarg.h>
#include <vector>
#include "../../univst_amd/csrc/fused.hip"
#pragma clang diagnostic ignored "-Wunused-result"

void uv_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
const char* uv_get_error() { return ""; }
void uv_prof_begin(int, double, double, hipStream_t, const char*) {}
void uv_prof_end(hipStream_t) {}

__global__ void fill(half_t* p, long n, unsigned seed, float scale) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned)i * 2654435761u + seed;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = (half_t)(((x & 0xffff) / 32768.f - 1.f) * scale);
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 196608, C = 320, heads = 8, T = 77, B = 3;
    half_t *X, *Wq, *Wo, *kv, *kvf, *Y, *bo;
    float *st, *wsum, *lnb, *st2;
    long long* tr;
    const long nb = (M + 63) / 64;
    hipMalloc(&X, (long)M * C * 2); hipMalloc(&Y, (long)M * C * 2); hipMalloc(&Wq, C * C * 2); hipMalloc(&Wo, C * C * 2); hipMalloc(&bo, C * 2);
    hipMalloc(&kv, B * T * 2 * C * 2); hipMalloc(&kvf, uv_attn2_kvf_halfs(B, heads, 40) * 2);
    hipMalloc(&st, (long)M * 4 * 4); hipMalloc(&st2, (long)M * 4 * 4); hipMalloc(&wsum, C * 4); hipMalloc(&lnb, C * 4); hipMalloc(&tr, nb * 4 * 8 * 8);
    fill<<<(unsigned)(((long)M * C + 255) / 256), 256>>>(X, (long)M * C, 1, 1.f);
    fill<<<(C * C + 255) / 256, 256>>>(Wq, C * C, 2, 0.05f);
    fill<<<(C * C + 255) / 256, 256>>>(Wo, C * C, 3, 0.05f);
    fill<<<(B * T * 2 * C + 255) / 256, 256>>>(kv, B * T * 2 * C, 4, 1.f);
    hipMemset(bo, 0, C * 2); hipMemset(wsum, 0, C * 4); hipMemset(lnb, 0, C * 4);
    std::vector<float> hst((long)M * 4);
    for (long i = 0; i < (long)M * 2; ++i) { hst[i * 2] = 0.f; hst[i * 2 + 1] = 160.f / 3.f; }
    hipMemcpy(st, hst.data(), (long)M * 16, hipMemcpyHostToDevice);
    uv_launch_kv_frag_pack(kv, kvf, B, T, C, heads, 0);
    Attn2Params p;
    p.X = X; p.ldx = C; p.M = M; p.ln_stats = st; p.ln_slots = 2; p.ln_wsum = wsum; p.ln_bias = lnb; p.Wq_f = Wq; p.kvf = kvf;
    p.rows_per_branch = (M / B) / 64 * 64; p.heads = heads; p.Nkv = T; p.q_prescaled = 1; p.Wo_f = Wo; p.bias_o = bo; p.R = X; p.ldr = C; p.Y = Y; p.ldy = C;
    p.stats_out = st2; p.trace = tr;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (uv_launch_attn2_fused(p, C, 0)) return 1;
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("launch %d: %.3f ms\n", rep, ms);
    }
    std::vector<long long> h(nb * 32);
    hipMemcpy(h.data(), tr, nb * 32 * 8, hipMemcpyDeviceToHost);
    const char* names[7] = {"X DMA + first weight frags -> barrier", "phase A k loop", "barrier + Q write (LN epilogue)", "-", "phase B (2 heads)", "phase C k loop (incl. its barrier wait)",
                            "residual wait + Y tile + row pass + stores"};
    const int a[7] = {0, 1, 2, 3, 3, 5, 6}, b[7] = {1, 2, 3, 3, 4, 6, 7};
    double sum[8] = {0}, tot = 0;
    for (long w = 0; w < nb * 4; ++w) {
        for (int i = 0; i < 7; ++i) sum[i] += (double)(h[w * 8 + b[i]] - h[w * 8 + a[i]]);
        tot += (double)(h[w * 8 + 7] - h[w * 8 + 0]);
    }
    // barrier wait before phase C: slot 4 -> 5
    double bw = 0;
    for (long w = 0; w < nb * 4; ++w) bw += (double)(h[w * 8 + 5] - h[w * 8 + 4]);
    printf("per wave, mean shader-clock cycles (s_memtime): total %.0f\n", tot / (nb * 4));
    for (int i = 0; i < 7; ++i)
        if (i != 3) printf("  %-50s %9.0f\n", names[i], sum[i] / (nb * 4));
    printf("  %-50s %9.0f\n", "  (of which: weight prefetch + barrier before phase C)", bw / (nb * 4));
    long long tmin = h[0], tmax = h[7];
    for (long w = 0; w < nb * 4; ++w) { if (h[w * 8] < tmin) tmin = h[w * 8]; if (h[w * 8 + 7] > tmax) tmax = h[w * 8 + 7]; }
    printf("first start -> last end: %lld ticks\n", tmax - tmin);
    return 0;
}
