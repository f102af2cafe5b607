// fp16 MFMA GEMM / implicit-GEMM convolution for gfx950 (CDNA4).
//
//   Y[m][n] = epilogue( sum_k X[m][k] * W[n][k] )          fp16 in, fp32 accumulate, fp16 out
//
// Both operands are K-contiguous ("B^T" layout), which is what v_mfma_f32_16x16x32_f16 wants: every lane
// feeds 8 consecutive k of one row.  The WEIGHT tile is the MFMA A operand and the ACTIVATION tile the B
// operand, so a lane's 4 accumulator registers are 4 consecutive output channels of one output row ->
// 8-byte stores and in-register bias / residual / GEGLU epilogues.
//
// MODE 0: X is a plain [M,K] matrix (Linear layers, attention projections, FF).
// MODE 1: X is gathered on the fly from NHWC activations (implicit GEMM): 3x3 pad-1 conv (stride 1/2),
//         1x1 conv, optional fused nearest x2 upsample of the input, optional channel-concat of two
//         sources (UNet skip connections) — none of these is ever materialised in HBM.
//         K index = tap * (C1 + C2) + c, weights pre-permuted to [Cout][ky][kx][Cin] at load time.
//
// Tile: 128 output rows x (NF*32) output channels x 64 k per step; 4 waves as 2(n) x 2(m), each wave
// NF x 4 fragments of 16x16.  Global -> registers -> LDS staging with the next tile's loads in flight
// under the MFMAs (guide T14); LDS rows padded to 160 B (conflict-free ds_read_b128, see common.h).
// Replaces (reference call sites): resnet.py:64 (PseudoConv3d 2-D conv), attention.py:123,141 (proj_in /
// proj_out), attention.py:375-377,425 + pnp_utils.py:39-43,97 (to_q/k/v/out), diffusers FeedForward/GEGLU,
// diffusers Attention projections, unet_3d_blocks.py:523,618 (torch.cat skip), resnet.py:145 (nearest x2).
#include "common.h"
#include "kernels.h"

namespace {

constexpr int BM = 128;
constexpr int BK = 64;
constexpr int LDSH = lds_stride_bytes(BK * 2) / 2;   // 80 halfs

template <int NF, int MODE>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmParams p) {
    constexpr int BN = NF * 32;
    __shared__ __attribute__((aligned(16))) half_t smem[(BM + BN) * LDSH];
    half_t* Xs = smem;
    half_t* Ws = smem + BM * LDSH;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wm = wave & 1;
    const int l15 = lane & 15, g = lane >> 4;

    const int nt_n = (p.N + BN - 1) / BN;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = lid % nt_n, tm = lid / nt_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- per-thread staging geometry: chunk kc (8 halfs) of rows rb + 32*i
    const int kc = tid & 7, rb = tid >> 3;
    const half_t* xptr[4];
    int x_iy0[4], x_ix0[4], x_img[4];
    bool x_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + rb + 32 * i;
        x_ok[i] = m < p.M;
        if (MODE == 0) {
            xptr[i] = p.X + (long)m * p.ldx + kc * 8;
        } else {
            int hw = p.Ho * p.Wo;
            int img = m / hw, rem = m - img * hw;
            int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            int pad = (p.taps == 9) ? 1 : 0;
            x_iy0[i] = oy * p.stride - pad;
            x_ix0[i] = ox * p.stride - pad;
            x_img[i] = img * p.Hs;
        }
    }
    const half_t* wptr[NF];
    bool w_ok[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        int n = n0 + rb + 32 * i;
        w_ok[i] = n < p.N;
        wptr[i] = p.W + (long)n * p.K + kc * 8;
    }
    const int Cin = p.C1 + p.C2;
    int tap = 0, cc = kc * 8;          // MODE 1: (tap, channel) of this thread's chunk in the current k tile
    if (MODE == 1) {
        while (cc >= Cin) { cc -= Cin; ++tap; }
    }

    h8 xr[4], wr[NF];
    const h8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    auto load_tile = [&](int k0) {
        const bool kok = (k0 + kc * 8) < p.K;
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                xr[i] = (x_ok[i] && kok) ? *reinterpret_cast<const h8*>(xptr[i] + k0) : zero8;
        } else {
            const int ky = (p.taps == 9) ? tap / 3 : 0;
            const int kx = (p.taps == 9) ? tap - 3 * ky : 0;
            const int He = p.Hs << p.up, We = p.Ws << p.up;
            const bool src2 = cc >= p.C1;
            const half_t* base = src2 ? p.X2 : p.X;
            const int cs = src2 ? p.C2 : p.C1;
            const int co = src2 ? cc - p.C1 : cc;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int iy = x_iy0[i] + ky, ix = x_ix0[i] + kx;
                bool ok = x_ok[i] && kok && iy >= 0 && iy < He && ix >= 0 && ix < We;
                long pix = (long)(x_img[i] + (iy >> p.up)) * p.Ws + (ix >> p.up);
                xr[i] = ok ? *reinterpret_cast<const h8*>(base + pix * cs + co) : zero8;
            }
        }
#pragma unroll
        for (int i = 0; i < NF; ++i)
            wr[i] = (w_ok[i] && kok) ? *reinterpret_cast<const h8*>(wptr[i] + k0) : zero8;
    };
    auto advance_k = [&]() {
        if (MODE == 1) {
            cc += BK;
            while (cc >= Cin) { cc -= Cin; ++tap; }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<h8*>(&Xs[(rb + 32 * i) * LDSH + kc * 8]) = xr[i];
#pragma unroll
        for (int i = 0; i < NF; ++i)
            *reinterpret_cast<h8*>(&Ws[(rb + 32 * i) * LDSH + kc * 8]) = wr[i];
    };

    f4 acc[NF][4];
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + BK - 1) / BK;
    load_tile(0);
    advance_k();
    store_tile();
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) {                 // next tile's global loads fly under this tile's MFMAs
            load_tile((kt + 1) * BK);
            advance_k();
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            h8 a[NF], b[4];
#pragma unroll
            for (int i = 0; i < NF; ++i)
                a[i] = *reinterpret_cast<const h8*>(&Ws[(wn * NF * 16 + i * 16 + l15) * LDSH + ks * 32 + g * 8]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                b[j] = *reinterpret_cast<const h8*>(&Xs[(wm * 64 + j * 16 + l15) * LDSH + ks * 32 + g * 8]);
#pragma unroll
            for (int i = 0; i < NF; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (kt + 1 < nk) {
            store_tile();
            __syncthreads();
        }
    }

    // ---- epilogue: lane holds rows n = nb + g*4 + r (r<4) of column m = mb + l15
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 64 + j * 16 + l15;
        if (m >= p.M) continue;
        const half_t* rbias = p.rowbias ? p.rowbias + (long)(m / p.rows_per_rb) * p.N : nullptr;
        if (p.geglu) {
            if constexpr (NF % 2 == 0) {
#pragma unroll
                for (int i = 0; i < NF; i += 2) {
                    const int nx = n0 + wn * NF * 16 + i * 16 + g * 4;        // x rows (permuted weight)
                    const int ng = nx + 16;                                    // gate rows
                    if (ng >= p.N) continue;
                    const int no = (n0 + wn * NF * 16) / 2 + (i / 2) * 16 + g * 4;
                    h4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float xv = acc[i][j][r] + (p.bias ? (float)p.bias[nx + r] : 0.f);
                        float gv = acc[i + 1][j][r] + (p.bias ? (float)p.bias[ng + r] : 0.f);
                        o[r] = (half_t)(xv * gelu_erf_f(gv));
                    }
                    *reinterpret_cast<h4*>(p.Y + (long)m * p.ldy + no) = o;
                }
            }
            continue;
        }
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int n = n0 + wn * NF * 16 + i * 16 + g * 4;
            if (n >= p.N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
            if (n + 3 < p.N) {
                if (p.bias) {
                    h4 bv = *reinterpret_cast<const h4*>(p.bias + n);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += (float)bv[r];
                }
                if (rbias) {
                    h4 bv = *reinterpret_cast<const h4*>(rbias + n);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += (float)bv[r];
                }
                if (p.R) {
                    h4 rv = *reinterpret_cast<const h4*>(p.R + (long)m * p.ldr + n);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
                }
                h4 o;
                if (p.bias2) {
                    h4 bv = *reinterpret_cast<const h4*>(p.bias2 + n);
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (half_t)((float)(half_t)v[r] + (float)bv[r]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (half_t)v[r];
                }
                *reinterpret_cast<h4*>(p.Y + (long)m * p.ldy + n) = o;
            } else {
                for (int r = 0; r < 4 && n + r < p.N; ++r) {
                    float t = v[r];
                    if (p.bias) t += (float)p.bias[n + r];
                    if (rbias) t += (float)rbias[n + r];
                    if (p.R) t += (float)p.R[(long)m * p.ldr + n + r];
                    half_t o = (half_t)t;
                    if (p.bias2) o = (half_t)((float)o + (float)p.bias2[n + r]);
                    p.Y[(long)m * p.ldy + n + r] = o;
                }
            }
        }
    }
}

// y[m][n] = sum_k act(x[m][k]) W[n][k] + b[n], M <= 8: one wave per output column (time embeddings).
__global__ __launch_bounds__(256) void linear_small_kernel(const half_t* __restrict__ x, const half_t* __restrict__ W,
                                                           const half_t* __restrict__ b, half_t* __restrict__ y,
                                                           int M, int N, int K, int silu_in) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float acc[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m] = 0.f;
    for (int k = lane * 8; k < K; k += 64 * 8) {
        h8 w = *reinterpret_cast<const h8*>(W + (long)n * K + k);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (m < M) {
                h8 xv = *reinterpret_cast<const h8*>(x + (long)m * K + k);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float xf = (float)xv[e];
                    if (silu_in) xf = silu_f(xf);
                    acc[m] += xf * (float)w[e];
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        float s = wave_sum(acc[m]);
        if (lane == 0 && m < M) y[(long)m * N + n] = (half_t)(s + (b ? (float)b[n] : 0.f));
    }
}

}  // namespace

int uv_launch_gemm(const GemmParams& p, int mode, hipStream_t stream) {
    UV_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
    UV_REQUIRE(p.K % 8 == 0, "gemm: K=%d must be a multiple of 8", p.K);
    UV_REQUIRE(p.N % 4 == 0 || true, "gemm: N");
    if (mode == 0) {
        UV_REQUIRE(p.ldx % 8 == 0, "gemm: ldx=%ld must be a multiple of 8", p.ldx);
    } else {
        UV_REQUIRE(p.C1 % 8 == 0 && p.C2 % 8 == 0, "conv: channel counts must be multiples of 8 (C1=%d C2=%d)", p.C1, p.C2);
        UV_REQUIRE(p.taps == 1 || p.taps == 9, "conv: taps=%d", p.taps);
        UV_REQUIRE(p.K == p.taps * (p.C1 + p.C2), "conv: K=%d != taps*(C1+C2)", p.K);
    }
    bool nf5 = !p.geglu && (p.N % 160 == 0) && (p.N % 128 != 0);
    int BN = nf5 ? 160 : 128;
    int nt = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    if (p.geglu) UV_REQUIRE(p.N % 32 == 0, "geglu: N=%d must be a multiple of 32", p.N);
    dim3 grid(nt), block(256);
    uv_prof_begin(mode == 0 ? UV_CLS_GEMM : UV_CLS_CONV, 2.0 * p.M * (double)p.N * p.K,
                  2.0 * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * (p.geglu ? p.N / 2 : p.N)), stream);
    if (mode == 0) {
        if (nf5) hipLaunchKernelGGL((gemm_kernel<5, 0>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((gemm_kernel<4, 0>), grid, block, 0, stream, p);
    } else {
        if (nf5) hipLaunchKernelGGL((gemm_kernel<5, 1>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((gemm_kernel<4, 1>), grid, block, 0, stream, p);
    }
    uv_prof_end(stream);
    UV_LAUNCH_CHECK();
    return UV_OK;
}

int uv_launch_linear_small(const half_t* x, const half_t* W, const half_t* b, half_t* y, int M, int N, int K,
                           int silu_in, hipStream_t stream) {
    UV_REQUIRE(M >= 1 && M <= 8, "linear_small: M=%d must be in 1..8", M);
    UV_REQUIRE(K % 8 == 0, "linear_small: K=%d must be a multiple of 8", K);
    hipLaunchKernelGGL(linear_small_kernel, dim3((N + 3) / 4), dim3(256), 0, stream, x, W, b, y, M, N, K, silu_in);
    UV_LAUNCH_CHECK();
    return UV_OK;
}
