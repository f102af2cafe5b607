// Common device/host helpers for the UniVST gfx950 kernels (CDNA4 only; no CUDA paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __fp16 fh4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

#define UV_OK 0
#define UV_ERR_ARG (-1)
#define UV_ERR_HIP (-2)
#define UV_ERR_UNSUPPORTED (-3)
#define UV_ERR_STATE (-4)

void uv_set_error(const char* fmt, ...);
const char* uv_get_error();

#define UV_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            uv_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));    \
            return UV_ERR_HIP;                                                                    \
        }                                                                                         \
    } while (0)

#define UV_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            uv_set_error(__VA_ARGS__);        \
            return UV_ERR_ARG;                \
        }                                     \
    } while (0)

#define UV_LAUNCH_CHECK()                                                                         \
    do {                                                                                          \
        hipError_t _e = hipGetLastError();                                                        \
        if (_e != hipSuccess) {                                                                   \
            uv_set_error("%s:%d: kernel launch -> %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
            return UV_ERR_HIP;                                                                    \
        }                                                                                         \
    } while (0)

// LDS row stride (bytes) that makes the 16-row x 4-chunk ds_read_b128 fragment pattern and the
// ds_read_b64_tr_b16 pattern conflict-free on gfx950: stride % 64 == 32 (brute-forced against the
// per-instruction lane groups of MI355X_MICROARCH.md §LDS).
constexpr int lds_stride_bytes(int row_bytes) {
    int s = (row_bytes + 63) / 64 * 64 + 32;
    return (s - 64 >= row_bytes) ? s - 64 : s;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }
// exact (erf) GELU with erf from Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far below fp16 resolution): one
// rcp + one exp2 + 8 fma/mul instead of the ~40-instruction libm erff — the GEGLU epilogue runs it 250 M times per
// layer.  gelu(x) = 0.5*x*(1 + erf(x/sqrt2)) = 0.5*(x + |x| * erf(|x|/sqrt2))  (erf is odd), so no sign select.
// GELU(tanh) of torch (approximate="tanh"): 0.5 x (1 + tanh(u)) = x * sigmoid(2u), u = sqrt(2/pi) (x + 0.044715 x^3): one exp2 + one rcp
__device__ __forceinline__ float gelu_tanh_f(float x) {
    const float u = 0.7978845608028654f * fmaf(0.044715f * x * x, x, x);
    return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(u * -2.8853900817779268f));
}

// (superseded in the GEGLU epilogues by geglu_erf2 below, round 4; kept as the scalar form the packed one was checked against)
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float ax = fabsf(x);
    const float z = ax * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(z * z * -1.4426950408889634f);
    const float pe = poly * t * e;                    // 1 - erf(z)
    return 0.5f * (x + fmaf(-ax, pe, ax));            // 0.5 * (x + |x| * (1 - pe))
}

// x * gelu_erf(g) for TWO values per call, transcendental-free (round 4).  The GEGLU epilogue of the FF1 projections evaluates 40 960
// GELUs per 256 x 320 tile with the matrix pipe idle (13.4 us of fixed cost per tile against 5.9 for a plain tile); the
// A&S form above costs ~19 issue slots per value (rcp and exp2 take two each and nothing in it packs).  Here
//     gelu(g) = 0.5 g (1 + S),  S = gc * Q(2 gc^2 / 25 - 1) ~= erf(gc / sqrt 2),  gc = clamp(g, -5, 5)
// with Q the degree-12 Chebyshev fit of erf(sqrt(u/2)) / sqrt(u) on u in [0, 25] (|S - erf| <= 7.3e-7 incl. the clamp: 1 - erf(5/sqrt 2) =
// 5.7e-7; |gelu error| <= 2e-6 absolute, three orders below fp16 resolution of the product that is stored).  Every operation is a
// v_pk_fma_f32 / v_pk_mul_f32 on the pair plus one v_med3_f32 per value: 10 issue slots per value (ISA-counted).
typedef float f2 __attribute__((ext_vector_type(2)));
// UV_GELU_DEG9 (A/B build, round 6: -DUV_GELU_DEG9): degree 9 with the clamp at 4.5 (|gelu error| <= 8.5e-5 absolute) instead of degree 12 at 5.0 (2e-6)
#ifdef UV_GELU_DEG9
#define UV_GELU_N 10
#define UV_GELU_CLAMP 4.5f
#define UV_GELU_S 0.09876543209876543f
#define UV_GELU_COEF {3.138136918e-01f, -1.543877219e-01f, 1.092108398e-01f, -8.021842318e-02f, 5.810019385e-02f, -3.804188117e-02f, 1.858792292e-02f, -1.050309624e-02f, 1.033601781e-02f, -4.680031871e-03f}
#else
#define UV_GELU_N 13
#define UV_GELU_CLAMP 5.f
#define UV_GELU_S 0.08f
#define UV_GELU_COEF {2.827276369e-01f, -1.405918177e-01f, 1.030358595e-01f, -8.090256480e-02f, 6.295351729e-02f, -4.642625655e-02f, 3.247217962e-02f, -2.261424982e-02f, 1.353305501e-02f, -5.053833458e-03f, 2.749192302e-03f, -3.353461957e-03f, 1.470752778e-03f}
#endif
__device__ __forceinline__ f2 geglu_erf2(f2 x, f2 g) {
    f2 gc;
    gc.x = __builtin_amdgcn_fmed3f(g.x, -UV_GELU_CLAMP, UV_GELU_CLAMP);
    gc.y = __builtin_amdgcn_fmed3f(g.y, -UV_GELU_CLAMP, UV_GELU_CLAMP);
    const f2 s = __builtin_elementwise_fma(gc * UV_GELU_S, gc, f2{-1.f, -1.f});
    constexpr float c[UV_GELU_N] = UV_GELU_COEF;
    f2 p = f2{c[UV_GELU_N - 1], c[UV_GELU_N - 1]};
#pragma unroll
    for (int i = UV_GELU_N - 2; i >= 0; --i) p = __builtin_elementwise_fma(p, s, f2{c[i], c[i]});
    const f2 S = gc * p;
    const f2 t = (x * 0.5f) * g;
    return __builtin_elementwise_fma(t, S, t);
}

// the same for two pairs with the two Horner chains interleaved link by link: a dependent v_pk_fma_f32 needs one wait state after its
// producer, and hipcc (at 245-251 VGPRs) schedules the chains of a block one after the other, padding every link of the second with an s_nop
__device__ __forceinline__ void geglu_erf2x2(f2 xa, f2 ga, f2 xb, f2 gb, f2& ya, f2& yb) {
    f2 ca, cb;
    ca.x = __builtin_amdgcn_fmed3f(ga.x, -UV_GELU_CLAMP, UV_GELU_CLAMP);
    ca.y = __builtin_amdgcn_fmed3f(ga.y, -UV_GELU_CLAMP, UV_GELU_CLAMP);
    cb.x = __builtin_amdgcn_fmed3f(gb.x, -UV_GELU_CLAMP, UV_GELU_CLAMP);
    cb.y = __builtin_amdgcn_fmed3f(gb.y, -UV_GELU_CLAMP, UV_GELU_CLAMP);
    const f2 sa = __builtin_elementwise_fma(ca * UV_GELU_S, ca, f2{-1.f, -1.f});
    const f2 sb = __builtin_elementwise_fma(cb * UV_GELU_S, cb, f2{-1.f, -1.f});
    constexpr float c[UV_GELU_N] = UV_GELU_COEF;
    f2 pa = f2{c[UV_GELU_N - 1], c[UV_GELU_N - 1]}, pb = f2{c[UV_GELU_N - 1], c[UV_GELU_N - 1]};
#pragma unroll
    for (int i = UV_GELU_N - 2; i >= 0; --i) {
        pa = __builtin_elementwise_fma(pa, sa, f2{c[i], c[i]});
        pb = __builtin_elementwise_fma(pb, sb, f2{c[i], c[i]});
        __builtin_amdgcn_sched_barrier(0);              // keep the two chains alternating
    }
    const f2 Sa = ca * pa, Sb = cb * pb;
    const f2 ta = (xa * 0.5f) * ga, tb = (xb * 0.5f) * gb;
    ya = __builtin_elementwise_fma(ta, Sa, ta);
    yb = __builtin_elementwise_fma(tb, Sb, tb);
}

// XCD-aware bijective block remap (8 XCDs, block b observed on XCD b%8): consecutive logical ids
// land on the same XCD so tiles that share an operand panel hit the same 4 MiB L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int nx = 8;
    if (nwg < nx) return bid;
    int q = nwg / nx, r = nwg % nx, xcd = bid % nx, idx = bid / nx;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
