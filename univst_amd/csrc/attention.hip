// Flash-style multi-source attention for gfx950: sparse-causal spatial self-attention (keys/values of
// each frame gathered BY POINTER from {prev, (cur), first} frames of its branch — never concatenated) and
// the 77-token text cross-attention, in one kernel.
//
// Per block: 4 waves x (QB*16) query rows of one (frame, head); K/V tiles of 64 keys staged in LDS.
// "Swapped" formulation (guide T12): S^T = K Q^T so a lane owns ONE query column -> the running max /
// sum are lane-local (+2 xor-shuffles across the 4 lane groups), the O^T accumulator is rescaled by a
// lane-uniform scalar, and exp'd probabilities feed the second MFMA straight from registers:
//     S^T[key][q]  = mfma_16x16x32( A = K[key][d..],  B = Q^T[d..][q] )
//     O^T[d][q]   += mfma_16x16x32( A = V^T[d][key..] (ds_read_b64_tr_b16 from row-major V), B = P^T[key..][q] )
// The k-slot <-> key permutation of the second MFMA is chosen so P^T needs no cross-lane movement.
// Softmax statistics in fp32, exp2 with the 1/sqrt(d)*log2(e) scale folded in.
// Replaces: attention.py:384-420 (SparseCausalAttention gather + SDPA), pnp_utils.py:59-92 (PnP gather +
// SDPA), diffusers AttnProcessor2_0 SDPA for attn2.
#include "common.h"
#include <stdlib.h>

#include "kernels.h"

namespace {

constexpr int KT = 64;   // keys per LDS tile
constexpr float DEFER = 8.f;   // deferred-rescale threshold in the exp2 domain (P <= 2^8 between rescales)

typedef __fp16 fh2 __attribute__((ext_vector_type(2)));

// 3-input max in ONE VALU op.  Written as inline asm because hipcc canonicalises (v_max_f32 x,x) every MFMA result
// before fmaxf() — 44 extra VALU instructions per key tile in this kernel (guide T17 / MI355X_MICROARCH "price of fillers").
// Inputs are never NaN here (finite scores or -inf masks), so IEEE NaN quieting is not needed.
__device__ __forceinline__ float max3f(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Two-phase attention (AttnParams::state_out / state_in): weights of the merge o = o1 * w1 + acc2 * w2, where o1 are the NORMALISED rows of the first
// phase with state (m1, l1) and acc2 the unnormalised accumulator of this launch with reference m2 and denominator l2 (log2 units).  l == 0 marks an
// empty phase.
__device__ __forceinline__ void attn_merge_coef(float m1, float l1, float m2, float l2, float& w1, float& w2) {
    const bool has1 = l1 > 0.f, has2 = l2 > 0.f;
    const float m = has1 ? (has2 ? fmaxf(m1, m2) : m1) : m2;
    const float a1 = has1 ? l1 * __builtin_amdgcn_exp2f(m1 - m) : 0.f;
    const float a2 = has2 ? __builtin_amdgcn_exp2f(m2 - m) : 0.f;
    const float den = a1 + l2 * a2;
    const float inv = den > 0.f ? 1.f / den : 0.f;
    w1 = a1 * inv;
    w2 = a2 * inv;
}
// a (frame, head, query block) without any key source in this phase (src_cnt == 0: e.g. the first local frame of a rank in the local phase of a PnP layer,
// whose two sources are both remote): phase 1 leaves zero rows with l = 0, phase 2 keeps what phase 1 wrote.  Block-uniform.
template <int RB>
__device__ __forceinline__ void attn_empty_phase(const AttnParams& p, int bf, int h, int qblk, int d) {
    if (!p.state_out) return;
    const int tid = threadIdx.x;
    const int dch = d / 4;
    for (int i = tid; i < RB * dch; i += 256) {
        const int r = i / dch, c = i - r * dch, qrow = qblk * RB + r;
        if (qrow < p.Nq) *reinterpret_cast<h4*>(p.o + ((long)bf * p.Nq + qrow) * p.ldo + h * d + c * 4) = h4{0, 0, 0, 0};
    }
    for (int r = tid; r < RB; r += 256) {
        const int qrow = qblk * RB + r;
        if (qrow < p.Nq) *reinterpret_cast<float2*>(p.state_out + (((long)bf * p.heads + h) * p.Nq + qrow) * 2) = make_float2(0.f, 0.f);
    }
}

// CF (round 5; prescaled q only, the wide heads without a spare V column): the running reference enters as the ACCUMULATOR of the first QK^T MFMA
// (cf[qb] = lw - mrun, what attn_pp64_kernel calls cfold), so the per-score scale fma disappears; the rare reference change shifts the pending scores.
// Without ONES the row sums are taken from the PACKED fp16 probabilities with v_dot2 (one VALU op per two keys, and the denominator is the sum of exactly
// the values the PV MFMA multiplies).  ISA count per 64 keys x 32 queries at head_dim 80: 198 -> ~150 VALU issue slots.
template <int DPAD, int DV16, int QB, bool CF = false, int TP = 0>
__device__ __forceinline__ void attn_body(const AttnParams& p) {
    constexpr int KSTR = lds_stride_bytes(DPAD * 2) / 2;
    constexpr int DV = DV16 * 16;
    constexpr int VSTR = lds_stride_bytes(DV * 2) / 2;
    constexpr int KS = DPAD / 32;
    constexpr int TILE = KT * KSTR + KT * VSTR;
    constexpr bool DBUF = (2 * TILE * 2) <= 48 * 1024;     // double-buffer when it keeps >= 3 blocks per CU
    // ONES: head_dim 40 leaves V columns 40..47 of the 48-wide V tile unused -> column 40 holds 1.0 so the PV MFMA
    // accumulates the softmax denominator (from the SAME fp16-rounded P that multiplies V) in a spare O^T row.
    constexpr bool ONES = (DV16 == 3);
    constexpr int D = ONES ? 40 : DV16 * 16;                 // the head dim this instantiation serves (checked by the dispatcher)
    __shared__ __attribute__((aligned(16))) half_t smem[(DBUF ? 2 : 1) * TILE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int nqb = (p.Nq + 64 * QB - 1) / (64 * QB);
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int qblk = lid % nqb;
    const int h = p.order ? lid / (nqb * p.BF) : (lid / nqb) % p.heads;        // head-major block order: see attn_pp40_kernel
    const int bf = p.order ? (lid / nqb) % p.BF : lid / (nqb * p.heads);
    constexpr int d = D;

    // ---- Q^T fragments (B operand): lane (q = l15, g) holds Q[q][ks*32 + g*8 .. +8]
    h8 qf[QB][KS];
    const h8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qrow = qblk * 64 * QB + wave * 16 * QB + qb * 16 + l15;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int dc = ks * 32 + g * 8;
            qf[qb][ks] = (qrow < p.Nq && dc < d)
                             ? *reinterpret_cast<const h8*>(p.q + ((long)bf * p.Nq + qrow) * p.ldq + h * d + dc)
                             : zero8;
        }
    }

    f4 o[DV16][QB];
    float mrun[QB], lrun[QB];      // running max in RAW score units; lrun unused when ONES
    float cf[QB];                  // CF: lw - mrun, the accumulator the scores start from
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        mrun[qb] = CF ? 0.f : -INFINITY;
        cf[qb] = 0.f;
        lrun[qb] = 0.f;
#pragma unroll
        for (int dv = 0; dv < DV16; ++dv) o[dv][qb] = f4{0.f, 0.f, 0.f, 0.f};
    }

    const int ntile = (p.Nkv + KT - 1) / KT;
    // Only the D real columns of a K / V row are staged per tile; the zero padding up to DPAD / DV (and, for ONES, the
    // 1.0 column) is written to both LDS buffers once, before the loop.
    // The prefetch loads are inline asm on purpose: hipcc's waitcnt insertion merges the "Q fragments pending" state of
    // the loop pre-header into every iteration and emitted s_waitcnt vmcnt(0) in front of the first QK^T MFMA, i.e. it
    // waited for the prefetch it had just issued (a full L2/HBM latency exposed per key tile).  With asm loads the
    // compiler tracks nothing; the single explicit vmcnt(0) sits in store_tile(), after the tile's MFMA + softmax work.
    constexpr int DCH = D / 8;                                // 16-byte chunks per row actually loaded
    constexpr int NL = (KT * DCH + 255) / 256;                // staging loads per thread, each for K and for V
    constexpr int REM = KT * DCH - (NL - 1) * 256;            // threads active in the last round (whole waves)
    static_assert(REM % 64 == 0, "staging rounds must be wave-uniform");
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    h8 kr[NL], vr[NL];
    unsigned gcol[NL], ksoff[NL], vsoff[NL];
    int srow[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int idx = tid + i * 256;
        const int row = idx / DCH, ch = idx - row * DCH;
        srow[i] = row;
        gcol[i] = (unsigned)(h * D + ch * 8);
        ksoff[i] = (unsigned)(row * KSTR + ch * 8);
        vsoff[i] = (unsigned)(KT * KSTR + row * VSTR + ch * 8);
    }
    auto gload16 = [](const half_t* ptr) {
        h8 v;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr));
        return v;
    };
    const int nsrc_eff = p.src_cnt ? p.src_cnt[bf] : p.nsrc;
    // the optional extra key segment (AttnParams::kx) follows the nsrc_eff equal-length sources as "source nsrc_eff" with its own
    // length, row stride and buffers
    const int ntile_x = p.kx ? (p.Nkv_x + KT - 1) / KT : 0;
    const int T = nsrc_eff * ntile + ntile_x;
    if (TP && T == 0) {                   // no key source in this phase (two-phase attention): block-uniform, before any barrier
        attn_empty_phase<64 * QB>(p, bf, h, qblk, d);
        return;
    }
    int nx_s = 0, nx_t = 0;               // (source, tile-in-source) of the NEXT tile to load
    long nx_off = (long)__builtin_amdgcn_readfirstlane(p.src_idx[bf * p.nsrc]) * p.Nkv * p.ldkv;      // element offset of that source's first key row
    bool ld_tail = false;                 // the tile in the prefetch registers has rows past Nkv (zeroed at store time)
    int ld_t0 = 0, ld_nkv = p.Nkv;
    auto load_tile = [&]() {              // global -> registers; completes under the MFMAs of the current tile
        const bool xs = nx_s >= nsrc_eff;                    // block-uniform: this tile belongs to the extra segment
        const int nkv = xs ? p.Nkv_x : p.Nkv;
        const long ld = xs ? p.ldkv_x : p.ldkv;
        const int t0 = nx_t * KT;
        const half_t* kb = (xs ? p.kx : p.k) + nx_off + (long)t0 * ld;
        const half_t* vb = (xs ? p.vx : p.v) + nx_off + (long)t0 * ld;
        ld_tail = t0 + KT > nkv;
        ld_t0 = t0;
        ld_nkv = nkv;
        const int rmax = nkv - 1 - t0;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            if (i + 1 < NL || wave_u * 64 < REM) {
                const int r = ld_tail ? (srow[i] < rmax ? srow[i] : rmax) : srow[i];      // clamp: never read past the source
                const unsigned off = (unsigned)(r * (int)ld) + gcol[i];
                kr[i] = gload16(kb + off);
                vr[i] = gload16(vb + off);
            }
        }
        if (++nx_t == (xs ? ntile_x : ntile)) {            // source boundary (<= nsrc times per block): fetch the next source frame index
            nx_t = 0;
            if (++nx_s < nsrc_eff) nx_off = (long)__builtin_amdgcn_readfirstlane(p.src_idx[bf * p.nsrc + nx_s]) * p.Nkv * p.ldkv;
            else if (nx_s == nsrc_eff && ntile_x) nx_off = (long)__builtin_amdgcn_readfirstlane(p.x_idx[bf]) * p.Nkv_x * p.ldkv_x;
        }
    };
    auto store_tile = [&](half_t* buf) {
        asm volatile("s_waitcnt vmcnt(0)");
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            if (i + 1 < NL || wave_u * 64 < REM) {
                asm volatile("" : "+v"(kr[i]), "+v"(vr[i]));          // the registers are defined from here on
                if (ld_tail && ld_t0 + srow[i] >= ld_nkv) { kr[i] = zero8; vr[i] = zero8; }
                *reinterpret_cast<h8*>(&buf[ksoff[i]]) = kr[i];
                *reinterpret_cast<h8*>(&buf[vsoff[i]]) = vr[i];
            }
        }
    };
    {   // one-time LDS image: zeros everywhere, 1.0 in V column D of every key row (ONES).  Rows past Nkv keep their 1.0:
        // their probabilities are exp2(-inf) = 0.
        constexpr int TOT = (DBUF ? 2 : 1) * TILE;
        for (int i = tid * 8; i < TOT; i += 256 * 8) *reinterpret_cast<h8*>(&smem[i]) = zero8;
        __syncthreads();
        if (ONES && tid < KT) {
            smem[KT * KSTR + tid * VSTR + D] = (half_t)1.f;
            if (DBUF) smem[TILE + KT * KSTR + tid * VSTR + D] = (half_t)1.f;
        }
    }

    const float c = p.q_prescaled ? 1.f : p.scale_log2e;
    load_tile();
    store_tile(smem);
    __syncthreads();

    int cur_s = 0, cur_t = 0;             // (source, tile-in-source) of the tile being consumed
    int cur_nkv = p.Nkv, cur_ntile = ntile;
    float lw = p.src_logw ? p.src_logw[bf * p.nsrc] : 0.f;            // log2 multiplicity of the current source ...
    float lwr = lw * (1.f / c);                                       // ... in raw-score units
    if (CF) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) cf[qb] = lw;
    }
    for (int tt = 0; tt < T; ++tt) {
        half_t* Ks = smem + (DBUF ? (tt & 1) * TILE : 0);
        half_t* Vs = Ks + KT * KSTR;
        if (tt + 1 < T) load_tile();
        const int t0 = cur_t * KT;

        // ---- S^T = K Q^T
        f4 sc[4][QB];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) sc[kb][qb] = CF ? f4{cf[qb], cf[qb], cf[qb], cf[qb]} : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                h8 a = *reinterpret_cast<const h8*>(&Ks[(kb * 16 + l15) * KSTR + ks * 32 + g * 8]);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb)
                    sc[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, qf[qb][ks], sc[kb][qb], 0, 0, 0);
            }
        }
        if (t0 + KT > cur_nkv) {          // tail tile: mask keys beyond the source's length (block-uniform branch)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (t0 + kb * 16 + g * 4 + r >= cur_nkv) sc[kb][qb][r] = -INFINITY;
        }
        h8 pb[QB][2];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            // row max over raw scores (lane-local over 16 values, then across the 4 lane groups)
            float mx = max3f(sc[0][qb][0], sc[0][qb][1], sc[0][qb][2]);
            mx = max3f(mx, sc[0][qb][3], sc[1][qb][0]);
            mx = max3f(mx, sc[1][qb][1], sc[1][qb][2]);
            mx = max3f(mx, sc[1][qb][3], sc[2][qb][0]);
            mx = max3f(mx, sc[2][qb][1], sc[2][qb][2]);
            mx = max3f(mx, sc[2][qb][3], sc[3][qb][0]);
            mx = max3f(mx, sc[3][qb][1], sc[3][qb][2]);
            mx = max3f(mx, sc[3][qb][3], sc[3][qb][3]);
            // deferred rescale (guide T13): keep the old running max while the new one is at most 2^DEFER larger in
            // the exp2 domain; P is then bounded by 2^DEFER (fp16 keeps full relative precision there) and the O^T
            // rescale + its accumulator traffic is skipped for the whole wave.  mrun is uniform over the 4 lane groups of a
            // query column, so "row max exceeds the bound" == "some lane's local max does": the cross-lane reduction (two
            // LDS-crossbar permutes + waits) is only paid inside the rare branch.
            if constexpr (CF) {
                // the scores already are s - mrun + lw: the deferred test is on them directly.  The first tile sets the reference whatever its
                // sign (a strongly negative first row max would underflow every probability); later changes only raise it.
                if (__builtin_amdgcn_ballot_w64(tt == 0 || mx > DEFER) != 0) {
                    const float o16 = __shfl_xor(mx, 16, 64);
                    mx = max3f(mx, o16, o16);
                    const float o32 = __shfl_xor(mx, 32, 64);
                    mx = max3f(mx, o32, o32);
                    float delta = tt == 0 ? mx : fmaxf(mx, 0.f);                    // row max (incl. the source's log2 multiplicity) above the current reference
                    if (!(delta > -3.0e38f)) delta = 0.f;                           // a fully masked first tile (-inf): keep the reference
                    mrun[qb] += delta;
                    cf[qb] = lw - mrun[qb];
                    const float alpha = tt == 0 ? 1.f : __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) {
                        sc[kb][qb][0] -= delta; sc[kb][qb][1] -= delta; sc[kb][qb][2] -= delta; sc[kb][qb][3] -= delta;
                    }
#pragma unroll
                    for (int dv = 0; dv < DV16; ++dv) {
                        o[dv][qb][0] *= alpha; o[dv][qb][1] *= alpha; o[dv][qb][2] *= alpha; o[dv][qb][3] *= alpha;
                    }
                    lrun[qb] *= alpha;
                }
                union { fh2 h[4]; h8 v; } u0, u1;
                const fh2 one2 = {(__fp16)1.f, (__fp16)1.f};
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    const float e0 = __builtin_amdgcn_exp2f(sc[kb][qb][0]), e1 = __builtin_amdgcn_exp2f(sc[kb][qb][1]);
                    const float e2 = __builtin_amdgcn_exp2f(sc[kb][qb][2]), e3 = __builtin_amdgcn_exp2f(sc[kb][qb][3]);
                    const fh2 lo = __builtin_amdgcn_cvt_pkrtz(e0, e1), hi = __builtin_amdgcn_cvt_pkrtz(e2, e3);
                    if (!ONES) {
                        lrun[qb] = __builtin_amdgcn_fdot2(lo, one2, lrun[qb], false);
                        lrun[qb] = __builtin_amdgcn_fdot2(hi, one2, lrun[qb], false);
                    }
                    if (kb < 2) { u0.h[(kb & 1) * 2] = lo; u0.h[(kb & 1) * 2 + 1] = hi; }
                    else { u1.h[(kb & 1) * 2] = lo; u1.h[(kb & 1) * 2 + 1] = hi; }
                }
                pb[qb][0] = u0.v;
                pb[qb][1] = u1.v;
                continue;
            }
            mx += lwr;
            float alpha = 1.f;
            if (__builtin_amdgcn_ballot_w64((mx - mrun[qb]) * c > DEFER) != 0) {
                const float o16 = __shfl_xor(mx, 16, 64);
                mx = max3f(mx, o16, o16);
                const float o32 = __shfl_xor(mx, 32, 64);
                mx = max3f(mx, o32, o32);
                const float mnew = fmaxf(mrun[qb], mx);
                alpha = __builtin_amdgcn_exp2f((mrun[qb] - mnew) * c);
                mrun[qb] = mnew;
#pragma unroll
                for (int dv = 0; dv < DV16; ++dv) {
                    o[dv][qb][0] *= alpha; o[dv][qb][1] *= alpha; o[dv][qb][2] *= alpha; o[dv][qb][3] *= alpha;
                }
                if (!ONES) lrun[qb] *= alpha;
            }
            const float mc = fmaf(-mrun[qb], c, lw);
            if (ONES) {
                union { fh2 h[4]; h8 v; } u0, u1;
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    float e0 = __builtin_amdgcn_exp2f(fmaf(sc[kb][qb][0], c, mc));
                    float e1 = __builtin_amdgcn_exp2f(fmaf(sc[kb][qb][1], c, mc));
                    float e2 = __builtin_amdgcn_exp2f(fmaf(sc[kb][qb][2], c, mc));
                    float e3 = __builtin_amdgcn_exp2f(fmaf(sc[kb][qb][3], c, mc));
                    fh2 lo = __builtin_amdgcn_cvt_pkrtz(e0, e1), hi = __builtin_amdgcn_cvt_pkrtz(e2, e3);
                    if (kb < 2) { u0.h[(kb & 1) * 2] = lo; u0.h[(kb & 1) * 2 + 1] = hi; }
                    else { u1.h[(kb & 1) * 2] = lo; u1.h[(kb & 1) * 2 + 1] = hi; }
                }
                pb[qb][0] = u0.v;
                pb[qb][1] = u1.v;
            } else {
                float rs = 0.f;
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float e = __builtin_amdgcn_exp2f(fmaf(sc[kb][qb][r], c, mc));
                        rs += e;
                        pb[qb][kb >> 1][(kb & 1) * 4 + r] = (half_t)e;
                    }
                lrun[qb] += rs;
            }
        }
        // ---- O^T += V^T P^T  (two 32-key chunks; V^T fragments shared by the QB query blocks)
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
#pragma unroll
            for (int dv = 0; dv < DV16; ++dv) {
                const half_t* vp = &Vs[(cc * 32 + g * 4 + (l15 >> 2)) * VSTR + dv * 16 + (l15 & 3) * 4];
                fh4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fh4*)(vp));
                fh4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fh4*)(vp + 16 * VSTR));
                h8 a;
                a[0] = (half_t)lo[0]; a[1] = (half_t)lo[1]; a[2] = (half_t)lo[2]; a[3] = (half_t)lo[3];
                a[4] = (half_t)hi[0]; a[5] = (half_t)hi[1]; a[6] = (half_t)hi[2]; a[7] = (half_t)hi[3];
#pragma unroll
                for (int qb = 0; qb < QB; ++qb)
                    o[dv][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pb[qb][cc], o[dv][qb], 0, 0, 0);
            }
        }
        if (!DBUF) __syncthreads();      // everyone finished reading the single buffer
        if (tt + 1 < T) store_tile(smem + (DBUF ? ((tt + 1) & 1) * TILE : 0));
        __syncthreads();
        if (++cur_t == cur_ntile) {
            cur_t = 0;
            if (++cur_s < nsrc_eff) {
                if (p.src_logw) {
                    lw = p.src_logw[bf * p.nsrc + cur_s];
                    lwr = lw * (1.f / c);
                }
            } else {                      // the extra segment: its own length, multiplicity 1
                cur_nkv = p.Nkv_x;
                cur_ntile = ntile_x;
                lw = 0.f;
                lwr = 0.f;
            }
            if (CF) {
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) cf[qb] = lw - mrun[qb];
            }
        }
    }

    // ---- finalize: O^T[d = dv*16 + g*4 + r][q = l15] / l
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        float l;
        if (ONES) {
            // denominator sits in O^T row d (= 40): register (d%16)%4 of lane group (d%16)/4, dv = d/16
            l = __shfl(o[DV16 - 1][qb][0], 32 + l15, 64);
        } else {
            l = lrun[qb];
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
        }
        float inv = 1.f / l, w1 = 0.f;
        const int qrow = qblk * 64 * QB + wave * 16 * QB + qb * 16 + l15;
        if (qrow >= p.Nq) continue;
        half_t* op = p.o + ((long)bf * p.Nq + qrow) * p.ldo + h * d;
        // two-phase attention: TP == 1 leaves the softmax state of this query (reference in log2 units, denominator), TP == 2 merges phase 1's rows with
        // this launch's accumulator.  Two instantiations on purpose: the first phase is the one-launch epilogue plus one store, and the merge weights exist
        // only where they are used (as one kernel with run-time branches the QB = 2 bodies at head_dim 40 / 80 produced sporadic x * w1 = +-0 elements).
        if constexpr (TP != 0) {
            const float m2 = CF ? mrun[qb] : mrun[qb] * c;
            const long srow = (((long)bf * p.heads + h) * p.Nq + qrow) * 2;
            if constexpr (TP == 1) {
                if (g == 0) *reinterpret_cast<float2*>(p.state_out + srow) = make_float2(m2, l);
            } else {
                const float2 st = *reinterpret_cast<const float2*>(p.state_in + srow);
                attn_merge_coef(st.x, st.y, m2, l, w1, inv);
            }
        }
#pragma unroll
        for (int dv = 0; dv < DV16; ++dv) {
            const int dc = dv * 16 + g * 4;
            if (dc < d) {
                h4 ov;
                if constexpr (TP == 2) {
                    const h4 o1 = *reinterpret_cast<const h4*>(op + dc);
#pragma unroll
                    for (int r = 0; r < 4; ++r) ov[r] = (half_t)fmaf((float)o1[r], w1, o[dv][qb][r] * inv);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) ov[r] = (half_t)(o[dv][qb][r] * inv);
                }
                *reinterpret_cast<h4*>(op + dc) = ov;
            }
        }
    }
}

template <int DPAD, int DV16, int QB, bool CF = false, int TP = 0>
__global__ __launch_bounds__(256) void attn_kernel(AttnParams p) {
    attn_body<DPAD, DV16, QB, CF, TP>(p);
}
// same body, register budget capped for 3 waves per SIMD (the small-head-dim kernels are VALU/latency bound:
// one more resident wave per SIMD hides the softmax behind another wave's MFMAs)
template <int DPAD, int DV16, int QB, bool CF = false, int TP = 0>
__global__ __launch_bounds__(256, 3) void attn_kernel_occ3(AttnParams p) {
    attn_body<DPAD, DV16, QB, CF, TP>(p);
}

// ----------------------------------------------------------------------------------------------------------------
// head_dim 40, long sequences (the 64x64-level self-attention: 5 launches per UNet call, a third of the step):
// a software-pipelined variant of attn_body<64, 3, 4>.
//
// Per key the softmax costs more VALU issue time (exp2, cvt, max) than the two MFMAs cost matrix time, and in the plain
// body a wave runs them back to back: QK^T, then softmax, then PV.  Here a wave works in 32-key steps with TWO score
// register sets: while the VALU exponentiates step s, the matrix pipe computes the scores of step s+1; while the matrix
// pipe runs PV of step s, the VALU takes the running max of step s+1 (guide T15, with the interleave pinned by
// sched_group_barrier so the in-order wave really alternates MFMA and VALU issue).  K/V tiles of 64 keys live in a 3-stage
// LDS ring (48 KB, two blocks per CU), prefetched two tiles ahead through registers; one barrier per tile.
//
// FOLD: the scale and the running max move INTO the QK^T MFMA.  Q arrives pre-multiplied by scale*log2(e) — the UNet graph
// folds that factor into the to_q weights (AttnParams::q_prescaled), so Q is rounded to fp16 once, exactly as often as the
// reference rounds it; multiplying an already rounded Q here would add a second 2^-11 error that shows on keys of very
// large norm (tests) — and the 24 padding columns of the 64-wide contraction carry, on the K side, three columns
// of 1.0 and, on the Q side, (-M_hi, -M_lo, +log2 multiplicity): the MFMA then returns s*c - M + lw directly and the
// per-score fma disappears (8 of ~29 VALU issue slots per query block and step).  M is the running reference quantised to
// 1/64 and split in a multiple of 8 plus a remainder so both parts are exact in fp16; softmax is shift invariant, so a
// quantised reference is not an approximation.  On a (rare) reference change the pending scores are shifted by the same
// delta and O^T is rescaled after the pending PV, the order T13 requires.
// TAG only names the symbol: 0 = self-attention, 1 = the 77-key text cross-attention of the same level (same code; separate rows
// in rocprofv3 --stats, so a per-symbol average means one kind of launch)
//
// STG = 1 (round 3): the K/V ring is filled by LDS-DMA (global_load_lds, 16 B per lane) instead of global -> registers -> ds_write.
// An ablation of the register-staged kernel (loads + stores removed, stale tiles) ran 21 % faster: per tile and wave 4 asm loads,
// an exposed vmcnt(0), 4 ds_write_b128 (13 cycles each, 24 % bank-conflicted) and their address arithmetic sat in a VALU-issue-bound
// loop.  The DMA writes lane-linear 16-byte pieces, so a stage is stored CHUNK-MAJOR: plane c = the 16-byte chunk c (8 head-dim
// columns) of all 64 key rows, 1 KB, filled by ONE wave instruction whose lane = key row.  K: 8 planes of 1024 B (0..4 data, 5 = the
// folded-reference constants (1, 1, 1, 0..), 6 / 7 zero — written once); the QK^T fragment of lane (key l15, g) is chunk ks*4+g of
// its row: plane stride 1024 B == 0 mod the 256-B bank span, so the rows of a ds_read_b128 lane group fall into disjoint banks.
// V: 6 planes (5 = the ones column) 1152 B apart — the two planes a ds_read_b64_tr_b16 group touches differ by 128 B mod 256 B.
// Ten DMA instructions per tile and BLOCK (2.5 per wave), no staging registers, and the only wait is the vmcnt(0) hipcc puts in
// front of the tile's barrier, a whole tile after the issue.  Rows past Nkv read a device zero page.
__device__ __attribute__((aligned(256))) half_t uv_attn_zero_page[128];

template <bool FOLD, int TAG = 0, int STG = 0, bool ONEB = true, bool K16 = false, int TP = 0>
__global__ __launch_bounds__(256, 2) void attn_pp40_kernel(AttnParams p) {
    static_assert(!K16 || FOLD, "the 16-wide second k step carries the folded reference columns");
    constexpr int NW = 4;
    constexpr int D = 40, DV16 = 3, QB = 4, NST = 3;
    constexpr int KSTR = lds_stride_bytes(64 * 2) / 2;       // 80 halfs
    constexpr int VSTR = lds_stride_bytes(48 * 2) / 2;       // 48 halfs
    constexpr int KPL = 512, VPL = 576;                      // STG: plane strides in halfs (1024 B / 1152 B)
    constexpr int KAREA = STG ? 8 * KPL : KT * KSTR;         // halfs of a stage taken by K
    constexpr int TILE = STG ? 8 * KPL + 6 * VPL : KT * KSTR + KT * VSTR;              // 14.75 KB / 16 KB per stage
    __shared__ __attribute__((aligned(16))) half_t smem[NST * TILE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int nqb = (p.Nq + 64 * NW - 1) / (64 * NW);
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    // Head-major order: consecutive logical blocks (= one XCD after xcd_remap) are the query blocks of ONE head over consecutive
    // frames.  Frame f+1 shares two of its three key sources with frame f (the clip's first frame and f itself), so their K/V
    // lines are still in that XCD's L2; with frames outermost every (frame, head) group re-fetched all of its sources from HBM
    // (PMC: 2.3x the compulsory bytes per launch).  AttnParams::order = 0 keeps the frame-major order (A/B aid UNIVST_ATTN_ORDER).
    const int qblk = lid % nqb;
    const int h = p.order ? lid / (nqb * p.BF) : (lid / nqb) % p.heads;
    const int bf = p.order ? (lid / nqb) % p.BF : lid / (nqb * p.heads);
    const float c = p.q_prescaled ? 1.f : p.scale_log2e;     // FOLD is only dispatched for prescaled q (c == 1)
    const h8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    const int ntile = (p.Nkv + KT - 1) / KT;
    const int nsrc_eff = p.src_cnt ? p.src_cnt[bf] : p.nsrc;
    const int T = nsrc_eff * ntile;
    if (TP && T == 0) {                   // no key source in this phase (two-phase attention): block-uniform, before any barrier
        attn_empty_phase<64 * NW>(p, bf, h, qblk, D);
        return;
    }
    float lw_cur = p.src_logw ? p.src_logw[bf * p.nsrc] : 0.f;

    // ---- Q^T fragments (B operand): lane (q = l15, g) holds Q[q][ks*32 + g*8 .. +8]
    h8 qf[QB][2];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qrow = qblk * 64 * NW + wave * 16 * QB + qb * 16 + l15;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int dc = ks * 32 + g * 8;
            h8 v = (qrow < p.Nq && dc < D) ? *reinterpret_cast<const h8*>(p.q + ((long)bf * p.Nq + qrow) * p.ldq + h * D + dc) : zero8;
            qf[qb][ks] = v;
        }
        if (FOLD && !K16 && g == 1) qf[qb][1][2] = (half_t)lw_cur;        // column 42: + log2 multiplicity of the source
        if (K16) {
            // K16: the second k step covers columns 32..47 only (head_dim 40 + the three folded-reference columns) on the 16-wide
            // MFMA: lane (q, g) holds columns 32 + 4g .. +3 in the LOW half of qf[qb][1] — g 0/1: q[32..39], g 2: (-M_hi, -M_lo, lw, 0)
            h8 v = zero8;
            if (qrow < p.Nq && g < 2) {
                const h4 t = *reinterpret_cast<const h4*>(p.q + ((long)bf * p.Nq + qrow) * p.ldq + h * D + 32 + g * 4);
                v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
            }
            if (g == 2) v[2] = (half_t)lw_cur;
            qf[qb][1] = v;
        }
    }
    // columns 40 / 41 of Q' <- -M (FOLD).  Returns the reference the MFMA will REALLY subtract: -(fp16(-hi) + fp16(-lo)), equal
    // to M whenever |M| < 16384 (hi a multiple of 8, lo a multiple of 1/64 below 8 are exact in fp16); beyond that the
    // bookkeeping follows the rounded value, so scores, O^T and the denominator stay on one common scale.
    auto set_shift = [&](int qb, float M) -> float {
        const float hi = floorf(M * 0.125f) * 8.f, lo = M - hi;
        const half_t hh = (half_t)(-hi), lh = (half_t)(-lo);
        if (g == (K16 ? 2 : 1)) {
            qf[qb][1][0] = hh;
            qf[qb][1][1] = lh;
        }
        return -((float)hh + (float)lh);
    };

    f4 o[DV16][QB];
    float mrun[QB], mc[QB];        // FOLD: mrun = quantised reference M (log2 units).  else: raw running max, mc = lw - mrun*c
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        mrun[qb] = FOLD ? 0.f : -INFINITY;
        mc[qb] = 0.f;
#pragma unroll
        for (int dv = 0; dv < DV16; ++dv) o[dv][qb] = f4{0.f, 0.f, 0.f, 0.f};
    }

    // ---- staging (see attn_body): 5 chunks of 16 B per K / V row, asm loads, explicit wait at store time
    constexpr int DCH = D / 8, NL = STG ? 1 : 2, REM = KT * DCH - 256;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    h8 kr[NL], vr[NL];
    unsigned gcol[NL], ksoff[NL], vsoff[NL];
    int srow[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int idx = tid + i * 256;
        const int row = idx / DCH, ch = idx - row * DCH;
        srow[i] = row;
        gcol[i] = (unsigned)(h * D + ch * 8);
        ksoff[i] = (unsigned)(row * KSTR + ch * 8);
        vsoff[i] = (unsigned)(KT * KSTR + row * VSTR + ch * 8);
    }
    auto gload16 = [](const half_t* ptr) {
        h8 v;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr));
        return v;
    };
    int ld_s = 0, ld_t = 0;
    long ld_off = (long)__builtin_amdgcn_readfirstlane(p.src_idx[bf * p.nsrc]) * p.Nkv * p.ldkv;
    bool ld_tail = false;
    int ld_t0 = 0;
    auto advance_ld = [&]() {
        if (++ld_t == ntile) {
            ld_t = 0;
            if (++ld_s < nsrc_eff) ld_off = (long)__builtin_amdgcn_readfirstlane(p.src_idx[bf * p.nsrc + ld_s]) * p.Nkv * p.ldkv;
        }
    };
    auto load_tile = [&]() {
        const int t0 = ld_t * KT;
        const half_t* kb = p.k + ld_off + (long)t0 * p.ldkv;
        const half_t* vb = p.v + ld_off + (long)t0 * p.ldkv;
        ld_tail = t0 + KT > p.Nkv;
        ld_t0 = t0;
        const int rmax = p.Nkv - 1 - t0;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            if (i + 1 < NL || wave_u * 64 < REM) {
                const int r = ld_tail ? (srow[i] < rmax ? srow[i] : rmax) : srow[i];
                const unsigned off = (unsigned)(r * (int)p.ldkv) + gcol[i];
                kr[i] = gload16(kb + off);
                vr[i] = gload16(vb + off);
            }
        }
        advance_ld();
    };
    auto store_tile = [&](half_t* buf) {
        asm volatile("s_waitcnt vmcnt(0)");
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            if (i + 1 < NL || wave_u * 64 < REM) {
                asm volatile("" : "+v"(kr[i]), "+v"(vr[i]));
                if (ld_tail && ld_t0 + srow[i] >= p.Nkv) { kr[i] = zero8; vr[i] = zero8; }
                *reinterpret_cast<h8*>(&buf[ksoff[i]]) = kr[i];
                *reinterpret_cast<h8*>(&buf[vsoff[i]]) = vr[i];
            }
        }
    };
    // STG: plane j of the tile (0..4 = K chunks, 5..9 = V chunks) is one DMA instruction of wave j % 4; lane = key row.
    // Inline asm on purpose (as the register prefetch above): for the builtin, hipcc's waitcnt pass assumes that the
    // ds_read_b64_tr_b16 intrinsic may alias the LDS-DMA in flight and puts s_waitcnt vmcnt(0) in front of the first V fragment
    // read of every tile, i.e. waits for the tile it has just requested.  M0 (the DMA's LDS base) is saved and restored inside
    // the statement; the single wait is the explicit vmcnt(0) in front of the tile's barrier.
    const half_t* const zpage = uv_attn_zero_page;
    const unsigned smem_lds = (unsigned)(size_t)((__attribute__((address_space(3))) half_t*)smem);      // LDS byte address of the ring
    auto glds16 = [](const half_t* src, unsigned lds_byte) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(lds_byte) : "memory");
    };
    auto dma_tile = [&](int stage_half) {                        // stage_half: offset of the stage in halfs from smem
        const int t0 = ld_t * KT;
        const bool rok = t0 + lane < p.Nkv;
        const long roff = ld_off + (long)(t0 + lane) * p.ldkv + h * D;
#pragma unroll
        for (int j = 0; j < (2 * DCH + NW - 1) / NW; ++j) {
            const int pl = wave_u + NW * j;                      // wave-uniform
            if ((j + 1) * NW <= 2 * DCH || pl < 2 * DCH) {
                const bool isv = pl >= DCH;
                const int c = isv ? pl - DCH : pl;
                const half_t* src = rok ? (isv ? p.v : p.k) + roff + c * 8 : zpage;
                const int dst = stage_half + (isv ? KAREA + c * VPL : c * KPL);
                glds16(src, __builtin_amdgcn_readfirstlane(smem_lds + 2u * (unsigned)dst));
            }
        }
        advance_ld();
    };
    {   // one-time LDS image: zeros, 1.0 in V column 40 (softmax denominator), FOLD: 1.0 in K columns 40..42
        for (int i = tid * 8; i < NST * TILE; i += NW * 64 * 8) *reinterpret_cast<h8*>(&smem[i]) = zero8;
        __syncthreads();
        if (tid < KT) {
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                if (STG) {          // chunk 5 of a row = columns 40..47: plane 5 of K and of V
                    smem[st * TILE + KAREA + 5 * VPL + tid * 8] = (half_t)1.f;
                    if (FOLD) {
                        smem[st * TILE + 5 * KPL + tid * 8] = (half_t)1.f;
                        smem[st * TILE + 5 * KPL + tid * 8 + 1] = (half_t)1.f;
                        smem[st * TILE + 5 * KPL + tid * 8 + 2] = (half_t)1.f;
                    }
                } else {
                    smem[st * TILE + KT * KSTR + tid * VSTR + D] = (half_t)1.f;
                    if (FOLD) {
                        smem[st * TILE + tid * KSTR + D] = (half_t)1.f;
                        smem[st * TILE + tid * KSTR + D + 1] = (half_t)1.f;
                        smem[st * TILE + tid * KSTR + D + 2] = (half_t)1.f;
                    }
                }
            }
        }
        if (STG) __syncthreads();          // the constants are in place before the first DMA is issued (no wave races ahead into a stage)
    }

    // ---- the four pipeline pieces
    const int kf_off = STG ? g * KPL + l15 * 8 : l15 * KSTR + g * 8;
    const int vf_off = STG ? KAREA + ((l15 & 3) >> 1) * VPL + (g * 4 + (l15 >> 2)) * 8 + (l15 & 1) * 4
                           : KT * KSTR + (g * 4 + (l15 >> 2)) * VSTR + (l15 & 3) * 4;
    auto kfrag_read = [&](const half_t* st, int hh, h8 (&kf)[2][2]) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (K16 && ks == 1) {       // columns 32 + 4g .. +3 of the key row: 8 bytes per lane
                    const int o16 = STG ? (4 + (g >> 1)) * KPL + l15 * 8 + (g & 1) * 4 + (hh * 32 + kb * 16) * 8
                                        : l15 * KSTR + 32 + g * 4 + (hh * 32 + kb * 16) * KSTR;
                    const h4 t = *reinterpret_cast<const h4*>(&st[o16]);
                    h8 v = zero8;
                    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
                    kf[kb][1] = v;
                } else {
                    kf[kb][ks] = *reinterpret_cast<const h8*>(&st[kf_off + (STG ? (hh * 32 + kb * 16) * 8 + ks * 4 * KPL : (hh * 32 + kb * 16) * KSTR + ks * 32)]);
                }
            }
    };
    auto vfrag_read = [&](const half_t* st, int hh, h8 (&vf)[DV16]) {
#pragma unroll
        for (int dv = 0; dv < DV16; ++dv) {
            const half_t* vp = &st[vf_off + (STG ? hh * 32 * 8 + dv * 2 * VPL : hh * 32 * VSTR + dv * 16)];
            fh4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fh4*)(vp));
            fh4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fh4*)(vp + 16 * (STG ? 8 : VSTR)));
            h8 a;
            a[0] = (half_t)lo[0]; a[1] = (half_t)lo[1]; a[2] = (half_t)lo[2]; a[3] = (half_t)lo[3];
            a[4] = (half_t)hi[0]; a[5] = (half_t)hi[1]; a[6] = (half_t)hi[2]; a[7] = (half_t)hi[3];
            vf[dv] = a;
        }
    };
    auto lo4 = [](const h8& v) { h4 r; r[0] = v[0]; r[1] = v[1]; r[2] = v[2]; r[3] = v[3]; return r; };
    // K16: the 16-wide step FIRST, from zero (half the matrix time of a 32-wide one), then the 32-wide step accumulates on it.
    // hipcc (ROCm 7.2) leaves too few wait states when a v_mfma_f32_16x16x32_f16 takes the result of a v_mfma_f32_16x16x16_f16 as
    // SrcC right behind it — the same toolchain hazard attn_text_kernel met, and it is its scheduler, not the source order, that decides
    // how close the two end up (a variant whose interleave hints let it pair them back to back returned wrong scores).  The two
    // two families are therefore separated by a data-flow fence (qk_fence) at every call site, and sit in separate scheduling regions:
    // whatever the scheduler does, a 32-wide MFMA reads an accumulator that the 16-wide one has finished writing
    // (tests/test_asm_hazards.py checks in the ISA that every such pair has the fence between them).
    auto qk_a = [&](const h8 (&kf)[2][2], f4 (&sc)[2][QB]) {
        const f4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) sc[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x16f16(lo4(kf[kb][1]), lo4(qf[qb][1]), z, 0, 0, 0);
    };
    // the fence between the two families: an empty asm that READS all eight 16-wide results and redefines the B operands of the
    // 32-wide MFMAs.  Every 16-wide MFMA must issue — and, because hipcc pads an MFMA result -> VGPR read correctly, must have WRITTEN
    // its result — before it; every 32-wide MFMA needs an operand defined by it and issues after it.
    auto qk_fence = [&](f4 (&sc)[2][QB]) {
        asm volatile("" : "+v"(qf[0][0]), "+v"(qf[1][0]), "+v"(qf[2][0]), "+v"(qf[3][0])
                     : "v"(sc[0][0]), "v"(sc[0][1]), "v"(sc[0][2]), "v"(sc[0][3]), "v"(sc[1][0]), "v"(sc[1][1]), "v"(sc[1][2]), "v"(sc[1][3]));
    };
    auto qk_b = [&](const h8 (&kf)[2][2], f4 (&sc)[2][QB]) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) sc[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kb][0], qf[qb][0], sc[kb][qb], 0, 0, 0);
    };
    auto qk = [&](const h8 (&kf)[2][2], f4 (&sc)[2][QB]) {        // S^T of one 32-key step: 16 MFMAs
        const f4 z = {0.f, 0.f, 0.f, 0.f};
        if (K16) {          // (prologue only: the loop interleaves the two halves with the softmax of the previous step)
            qk_a(kf, sc);
            qk_fence(sc);
            __builtin_amdgcn_sched_barrier(0);
            qk_b(kf, sc);
            return;
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) sc[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kb][0], qf[qb][0], z, 0, 0, 0);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) sc[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kb][1], qf[qb][1], sc[kb][qb], 0, 0, 0);
    };
    auto exp_part = [&](const f4 (&sc)[2][QB], h8 (&pb)[QB]) {    // P^T (fp16, B operand of the PV MFMA)
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            union { fh2 h[4]; h8 v; } u;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                float e0, e1, e2, e3;
                if (FOLD) {
                    e0 = __builtin_amdgcn_exp2f(sc[kb][qb][0]); e1 = __builtin_amdgcn_exp2f(sc[kb][qb][1]);
                    e2 = __builtin_amdgcn_exp2f(sc[kb][qb][2]); e3 = __builtin_amdgcn_exp2f(sc[kb][qb][3]);
                } else {
                    e0 = __builtin_amdgcn_exp2f(fmaf(sc[kb][qb][0], c, mc[qb])); e1 = __builtin_amdgcn_exp2f(fmaf(sc[kb][qb][1], c, mc[qb]));
                    e2 = __builtin_amdgcn_exp2f(fmaf(sc[kb][qb][2], c, mc[qb])); e3 = __builtin_amdgcn_exp2f(fmaf(sc[kb][qb][3], c, mc[qb]));
                }
                u.h[kb * 2] = __builtin_amdgcn_cvt_pkrtz(e0, e1);
                u.h[kb * 2 + 1] = __builtin_amdgcn_cvt_pkrtz(e2, e3);
            }
            pb[qb] = u.v;
        }
    };
    auto exp_half = [&](const f4 (&sc)[2][QB], h8 (&pb)[QB], int half) {      // query blocks 2*half, 2*half + 1 (FOLD arithmetic)
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
            const int qb = half * 2 + q2;
            union { fh2 h[4]; h8 v; } u;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const float e0 = __builtin_amdgcn_exp2f(sc[kb][qb][0]), e1 = __builtin_amdgcn_exp2f(sc[kb][qb][1]);
                const float e2 = __builtin_amdgcn_exp2f(sc[kb][qb][2]), e3 = __builtin_amdgcn_exp2f(sc[kb][qb][3]);
                u.h[kb * 2] = __builtin_amdgcn_cvt_pkrtz(e0, e1);
                u.h[kb * 2 + 1] = __builtin_amdgcn_cvt_pkrtz(e2, e3);
            }
            pb[qb] = u.v;
        }
    };
    auto pv = [&](const h8 (&vf)[DV16], const h8 (&pb)[QB]) {      // O^T += V^T P^T: 12 MFMAs
#pragma unroll
        for (int dv = 0; dv < DV16; ++dv)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) o[dv][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[dv], pb[qb], o[dv][qb], 0, 0, 0);
    };
    auto local_max = [&](const f4 (&sc)[2][QB], float (&mx)[QB]) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            float m = max3f(sc[0][qb][0], sc[0][qb][1], sc[0][qb][2]);
            m = max3f(m, sc[0][qb][3], sc[1][qb][0]);
            m = max3f(m, sc[1][qb][1], sc[1][qb][2]);
            mx[qb] = max3f(m, sc[1][qb][3], sc[1][qb][3]);
        }
    };
    auto mask_tail = [&](f4 (&sc)[2][QB], int key0) {              // keys >= Nkv of a tail tile -> -inf
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (key0 + kb * 16 + g * 4 + r >= p.Nkv) sc[kb][qb][r] = -INFINITY;
    };
    // reference update for the scores `sc` of the NEXT step (deferred, T13).  Everything exponentiated so far is already in
    // O^T (the PV of the previous step precedes this in program order), so O^T is rescaled exactly once per change.
    // ONEB: one wave-wide test over the four query blocks in front of the per-block tests (the common case — no reference moves —
    // then costs 1 max3 + 1 max + 1 compare + 1 branch instead of 4 compares + 4 branches: +1.2 % in four same-box A/B pairs)
    auto decide = [&](f4 (&sc)[2][QB], const float (&mx)[QB], float lw, bool first) {
        if (FOLD && ONEB && !first) {
            const float mall = fmaxf(max3f(mx[0], mx[1], mx[2]), mx[3]);
            if (__builtin_amdgcn_ballot_w64(mall > DEFER) == 0) return;
        }
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            if (FOLD) {
                if (__builtin_amdgcn_ballot_w64(first || mx[qb] > DEFER) != 0) {
                    float m = mx[qb];
                    const float o16 = __shfl_xor(m, 16, 64);
                    m = max3f(m, o16, o16);
                    const float o32 = __shfl_xor(m, 32, 64);
                    m = max3f(m, o32, o32);
                    float delta = floorf(m * 64.f + 0.5f) * (1.f / 64.f);          // shifted row max, quantised
                    if (!first) delta = fmaxf(delta, 0.f);
                    const float mnew = set_shift(qb, mrun[qb] + delta);
                    delta = mnew - mrun[qb];
                    mrun[qb] = mnew;
                    // (first reference: O^T is still zero, and 2^-delta overflows for a strongly negative first row max — 0 * inf)
                    const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        sc[kb][qb][0] -= delta; sc[kb][qb][1] -= delta; sc[kb][qb][2] -= delta; sc[kb][qb][3] -= delta;
                    }
#pragma unroll
                    for (int dv = 0; dv < DV16; ++dv) {
                        o[dv][qb][0] *= alpha; o[dv][qb][1] *= alpha; o[dv][qb][2] *= alpha; o[dv][qb][3] *= alpha;
                    }
                }
            } else {
                const float lwr = lw * (1.f / c);
                float m = mx[qb] + lwr;
                if (__builtin_amdgcn_ballot_w64((m - mrun[qb]) * c > DEFER) != 0) {
                    const float o16 = __shfl_xor(m, 16, 64);
                    m = max3f(m, o16, o16);
                    const float o32 = __shfl_xor(m, 32, 64);
                    m = max3f(m, o32, o32);
                    const float mnew = fmaxf(mrun[qb], m);
                    const float alpha = __builtin_amdgcn_exp2f((mrun[qb] - mnew) * c);
                    mrun[qb] = mnew;
#pragma unroll
                    for (int dv = 0; dv < DV16; ++dv) {
                        o[dv][qb][0] *= alpha; o[dv][qb][1] *= alpha; o[dv][qb][2] *= alpha; o[dv][qb][3] *= alpha;
                    }
                }
                mc[qb] = fmaf(-mrun[qb], c, lw);
            }
        }
    };
    auto set_lw = [&](float lw) {                                  // FOLD: column 42 of Q' <- log2 multiplicity of the source
        if (g == (K16 ? 2 : 1)) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) qf[qb][1][2] = (half_t)lw;
        }
    };
// pin the issue order inside the two overlapped regions (LLVM SchedGroupMask: VALU 0x2, MFMA 0x8, DS read 0x100, TRANS 0x400)
#define UV_PP_PHASE1()                                                            \
    _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) {                           \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        \
        if (i_ < 6) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);            \
        if (!FOLD) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);             \
        __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);                        \
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                        \
    }
/* K16: a step's scores come in two scheduling regions (8 MFMAs each), each with half of the previous step's exponentials */ \
#define UV_PP_PHASE1H(DS)                                                         \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                            \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        \
        if ((DS) && i_ < 6) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    \
        __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);                        \
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                        \
    }
#define UV_PP_PHASE2()                                                            \
    _Pragma("unroll") for (int i_ = 0; i_ < 12; ++i_) {                           \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        \
        if (i_ < 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);            \
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                        \
    }

// an empty asm that "uses" four values: keeps their producers in this basic block (LLVM otherwise sinks the exp2 / max
// work below the next branch, out of reach of the interleave above)
#define UV_PP_PIN4(a) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]))
#define UV_PP_PIN2(a, i) asm volatile("" : "+v"(a[i]), "+v"(a[(i) + 1]))

    // ---- prologue: tiles 0 and 1 into the ring, scores + reference of step (0, 0)
    if (STG) {
        dma_tile(0);
        if (T > 1) dma_tile(TILE);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        load_tile();
        store_tile(smem);
        if (T > 1) {
            load_tile();
            store_tile(smem + TILE);
        }
    }
    __syncthreads();

    f4 scA[2][QB], scB[2][QB];
    h8 kf[2][2], vf[DV16], pb[QB];
    float mx[QB];
    int nx_s = 0, nx_t = (ntile > 1) ? 1 : 0;            // (source, tile-in-source) of tile tt+1
    if (ntile == 1) nx_s = 1;
    float lw_nxt = (nx_s != 0 && nx_s < nsrc_eff && p.src_logw) ? p.src_logw[bf * p.nsrc + nx_s] : lw_cur;
    int t0_cur = 0, t0_nxt = nx_t * KT;
    half_t* b_cur = smem;
    half_t* b_nxt = smem + TILE;
    half_t* b_ld = smem + 2 * TILE;

    kfrag_read(b_cur, 0, kf);
    qk(kf, scA);
    if (t0_cur + KT > p.Nkv) mask_tail(scA, t0_cur);
    local_max(scA, mx);
    decide(scA, mx, lw_cur, true);
    kfrag_read(b_cur, 1, kf);

    for (int tt = 0; tt < T; ++tt) {
        const bool has_next = tt + 1 < T;
        if (tt + 2 < T) {                                // tile tt+2, two tiles ahead: -> registers, or (STG) straight into the stage
            if (STG) dma_tile((int)(b_ld - smem));       // tile tt-1 left at the last barrier; lands before this iteration's barrier
            else load_tile();
        }
        // ---- step (tt, 0): scores of (tt, 1) on the matrix pipe while (tt, 0) is exponentiated
        vfrag_read(b_cur, 0, vf);
        if (K16) {
            qk_a(kf, scB);
            exp_half(scA, pb, 0);
            UV_PP_PIN2(pb, 0);
            UV_PP_PHASE1H(1);
            qk_fence(scB);
            __builtin_amdgcn_sched_barrier(0);
            qk_b(kf, scB);
            exp_half(scA, pb, 1);
            UV_PP_PIN2(pb, 2);
            UV_PP_PHASE1H(0);
        } else {
            qk(kf, scB);
            exp_part(scA, pb);
            UV_PP_PIN4(pb);
            UV_PP_PHASE1();
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t0_cur + KT > p.Nkv) mask_tail(scB, t0_cur + 32);
        kfrag_read(b_nxt, 0, kf);
        pv(vf, pb);
        local_max(scB, mx);
        UV_PP_PIN4(mx);
        UV_PP_PHASE2();
        __builtin_amdgcn_sched_barrier(0);
        decide(scB, mx, lw_cur, false);
        // ---- step (tt, 1): scores of (tt+1, 0)
        if (FOLD && lw_nxt != lw_cur) set_lw(lw_nxt);
        vfrag_read(b_cur, 1, vf);
        if (K16) {
            qk_a(kf, scA);
            exp_half(scB, pb, 0);
            UV_PP_PIN2(pb, 0);
            UV_PP_PHASE1H(1);
            qk_fence(scA);
            __builtin_amdgcn_sched_barrier(0);
            qk_b(kf, scA);
            exp_half(scB, pb, 1);
            UV_PP_PIN2(pb, 2);
            UV_PP_PHASE1H(0);
        } else {
            qk(kf, scA);
            exp_part(scB, pb);
            UV_PP_PIN4(pb);
            UV_PP_PHASE1();
        }
        __builtin_amdgcn_sched_barrier(0);
        if (has_next && t0_nxt + KT > p.Nkv) mask_tail(scA, t0_nxt);
        kfrag_read(b_nxt, 1, kf);
        pv(vf, pb);
        local_max(scA, mx);
        UV_PP_PIN4(mx);
        UV_PP_PHASE2();
        __builtin_amdgcn_sched_barrier(0);
        if (has_next) decide(scA, mx, lw_nxt, false);
        // ---- end of tile: tile tt+2 into the stage tile tt-1 vacated one barrier ago
        if (!STG && tt + 2 < T) store_tile(b_ld);
        if (STG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of tile tt+2 have landed
        __syncthreads();
        half_t* tb = b_cur; b_cur = b_nxt; b_nxt = b_ld; b_ld = tb;
        t0_cur = t0_nxt;
        lw_cur = lw_nxt;
        if (++nx_t == ntile) {
            nx_t = 0;
            if (++nx_s < nsrc_eff && p.src_logw) lw_nxt = p.src_logw[bf * p.nsrc + nx_s];
        }
        t0_nxt = nx_t * KT;
    }
#undef UV_PP_PHASE1
#undef UV_PP_PHASE1H
#undef UV_PP_PIN2
#undef UV_PP_PIN4
#undef UV_PP_PHASE2

    // ---- finalize: O^T[d = dv*16 + g*4 + r][q = l15] / l, l = O^T row 40 (the ones column of V)
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const float l = __shfl(o[DV16 - 1][qb][0], 32 + l15, 64);
        float inv = 1.f / l, w1 = 0.f;
        const int qrow = qblk * 64 * NW + wave * 16 * QB + qb * 16 + l15;
        if (qrow >= p.Nq) continue;
        half_t* op = p.o + ((long)bf * p.Nq + qrow) * p.ldo + h * D;
        if constexpr (TP != 0) {          // two-phase attention (AttnParams; TP == 1 first phase, 2 merge phase): the reference in log2 units is M (FOLD) or the raw running max times c
            const float m2 = FOLD ? mrun[qb] : mrun[qb] * c;
            const long srow = (((long)bf * p.heads + h) * p.Nq + qrow) * 2;
            if constexpr (TP == 1) {
                if (g == 0) *reinterpret_cast<float2*>(p.state_out + srow) = make_float2(m2, l);
            } else {
                const float2 st = *reinterpret_cast<const float2*>(p.state_in + srow);
                attn_merge_coef(st.x, st.y, m2, l, w1, inv);
            }
        }
#pragma unroll
        for (int dv = 0; dv < DV16; ++dv) {
            const int dc = dv * 16 + g * 4;
            if (dc < D) {
                h4 ov;
                if constexpr (TP == 2) {
                    const h4 o1 = *reinterpret_cast<const h4*>(op + dc);
#pragma unroll
                    for (int r = 0; r < 4; ++r) ov[r] = (half_t)fmaf((float)o1[r], w1, o[dv][qb][r] * inv);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) ov[r] = (half_t)(o[dv][qb][r] * inv);
                }
                *reinterpret_cast<h4*>(op + dc) = ov;
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------
// head_dim 64, long sequences (round 3): the joint attention of the SD3 / SD3.5 MM-DiT (4096 image + 333 text tokens per frame at
// 1024 px, 62 % of that step in attn_body) and the SD-v2.1 layout.  Same software pipeline as attn_pp40_kernel — two score register
// sets, QK^T of step s+1 on the matrix pipe under the exponentials of step s, PV of step s under the row maxima of step s+1, K/V
// tiles of 64 keys in a 3-stage LDS ring filled by LDS-DMA in chunk-major planes — with what head_dim 64 changes:
//   * no padding columns to fold the reference into: Q arrives prescaled (AttnParams::q_prescaled) and the running reference enters
//     as the ACCUMULATOR of the first QK^T MFMA: cfold[qb] = (lw - M) in all four slots, a persistent fp32 register quad per query
//     block that changes only when the reference or the source multiplicity does.  The MFMA returns s - M + lw, the exponentials
//     take it as is; one MFMA family in the loop, so no mixed-family fence (attn_pp40_kernel K16);
//   * no spare V column for the denominator either: the row sums are taken from the PACKED fp16 probabilities with
//     v_dot2c_f32_f16 against (1, 1) — one VALU op per two keys, and the denominator is the sum of exactly the values the PV MFMA
//     multiplies (what the ones column gave the d = 40 kernel);
//   * 16 data planes per tile and stage (8 K + 8 V), four DMA instructions per wave and tile, no constant planes, no LDS init;
//   * the EXTRA key segment of the joint attention (AttnParams::kx: the frame's text tokens, own length / stride / buffers,
//     multiplicity 1) is one more source of the tile sequence.
// D, QB (round 5): the same kernel at head_dim 80 (the 32x32 level of the SD-v1.5 UNet) with QB = 2 query blocks (32 rows) per wave: three 32-wide k
// steps whose last half is padding (two K planes of a stage that are zeroed once and never written by the DMAs; Q is zero there), five V^T
// fragments, 20 data planes per tile = five DMA instructions per wave; 23.25 KB per stage, 70 KB per block, two blocks per CU.
// Register budget (2 waves per SIMD: 256): O^T 64 + two score sets 64 + Q 32 + cfold 16 + K / V / P fragments 48 + bookkeeping: 256, no
// spill.  Measured (12 frames x 24 heads x 4096 queries over 3 x 4096 keys, same box): 3.85 ms = 963 TF against 4.23 ms = 878 TF of
// attn_body<64, 4, 4>; with ONE wave per SIMD 6.79 ms (546 TF: the second wave hides the tile barrier and the LDS latency); without
// the sched_group_barrier pins 3.91 ms, with the conversions issued one MFMA later 3.85 ms — the interleave does not matter, because
// per SIMD the step costs the SUM of its MFMA cycles (32 x 16 = 512) and its VALU / transcendental issue cycles (~450): the two do not
// overlap across the two waves of a SIMD (tools/probes/coissue_probe.hip) and hardly inside one here.  What is left is less work per
// key, not a better order.
template <int D = 64, int QB = 4, int TAG = 0, int TP = 0>
__global__ __launch_bounds__(256, 2) void attn_pp64_kernel(AttnParams p) {
    static_assert(D % 16 == 0 && D % 8 == 0 && (QB == 2 || QB == 4), "whole V^T fragments, 16-byte K / V chunks");
    constexpr int NW = 4, DV16 = D / 16, NST = 3;
    constexpr int KS = (D + 31) / 32;                        // 32-wide k steps of QK^T (D = 80: the third is half padding)
    constexpr int NDK = D / 8, NPK = KS * 4, NPV = D / 8;    // data chunks of a K row, K planes of a stage (NPK - NDK zero planes), V planes
    constexpr int RB = NW * 16 * QB;                         // query rows of a block
    constexpr int KPL = 512, VPL = 576;                      // plane strides in halfs (1024 B / 1152 B: see attn_pp40_kernel STG)
    constexpr int KAREA = NPK * KPL;
    constexpr int TILE = NPK * KPL + NPV * VPL;              // 17 KB per stage (D = 64), 23.25 KB (D = 80)
    __shared__ __attribute__((aligned(16))) half_t smem[NST * TILE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int nqb = (p.Nq + RB - 1) / RB;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int qblk = lid % nqb;
    const int h = p.order ? lid / (nqb * p.BF) : (lid / nqb) % p.heads;
    const int bf = p.order ? (lid / nqb) % p.BF : lid / (nqb * p.heads);
    const h8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    const int ntile = (p.Nkv + KT - 1) / KT;
    const int ntile_x = p.kx ? (p.Nkv_x + KT - 1) / KT : 0;
    const int nsrc_eff = p.src_cnt ? p.src_cnt[bf] : p.nsrc;
    const int nseg = nsrc_eff + (p.kx ? 1 : 0);
    const int T = nsrc_eff * ntile + ntile_x;
    if (TP && T == 0) {                   // no key source in this phase (two-phase attention, TP = 1 first phase / 2 merge phase): block-uniform, before any barrier
        attn_empty_phase<NW * 16 * QB>(p, bf, h, qblk, D);
        return;
    }
    auto seg_lw = [&](int sidx) { return (sidx < nsrc_eff && p.src_logw) ? p.src_logw[bf * p.nsrc + sidx] : 0.f; };
    auto seg_nkv = [&](int sidx) { return sidx < nsrc_eff ? p.Nkv : p.Nkv_x; };
    auto seg_ntile = [&](int sidx) { return sidx < nsrc_eff ? ntile : ntile_x; };

    // ---- Q^T fragments (B operand): lane (q = l15, g) holds Q[q][ks*32 + g*8 .. +8]
    h8 qf[QB][KS];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qrow = qblk * RB + wave * 16 * QB + qb * 16 + l15;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            qf[qb][ks] = (qrow < p.Nq && ks * 32 + g * 8 < D) ? *reinterpret_cast<const h8*>(p.q + ((long)bf * p.Nq + qrow) * p.ldq + h * D + ks * 32 + g * 8) : zero8;
    }

    f4 o[DV16][QB], cfold[QB];
    float mrun[QB], lsum[QB];
    float lw_cur = seg_lw(0);
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        mrun[qb] = 0.f;
        lsum[qb] = 0.f;
        cfold[qb] = f4{lw_cur, lw_cur, lw_cur, lw_cur};
#pragma unroll
        for (int dv = 0; dv < DV16; ++dv) o[dv][qb] = f4{0.f, 0.f, 0.f, 0.f};
    }

    // ---- K/V ring: plane pl (0..7 K chunks, 8..15 V chunks) of a tile is one DMA instruction of wave pl % 4; lane = key row
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const half_t* const zpage = uv_attn_zero_page;
    const unsigned smem_lds = (unsigned)(size_t)((__attribute__((address_space(3))) half_t*)smem);
    auto glds16 = [](const half_t* src, unsigned lds_byte) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(lds_byte) : "memory");
    };
    // loader state: plain integers only (element offsets from p.k / p.v — the extra segment's buffers are addressed through the
    // pointer DIFFERENCE kx - k, vx - v).  With pointer-typed state set inside a nested lambda hipcc kept the state in a stack object
    // and read it back through FLAT loads, which faulted.
    const long xk_delta = p.kx ? (long)(p.kx - p.k) : 0, xv_delta = p.kx ? (long)(p.vx - p.v) : 0;
    int ld_s = 0, ld_t = 0;
    long ld_koff, ld_voff, ld_ld;
    int ld_nkv, ld_ntile;
    {
        const long off = (long)__builtin_amdgcn_readfirstlane(p.src_idx[bf * p.nsrc]) * p.Nkv * p.ldkv + h * D;
        ld_koff = off; ld_voff = off; ld_ld = p.ldkv; ld_nkv = p.Nkv; ld_ntile = ntile;
    }
    auto dma_tile = [&](int stage_half) {
        const int t0 = ld_t * KT;
        const bool rok = t0 + lane < ld_nkv;
        const long roff = (long)(t0 + lane) * ld_ld;
        const half_t* const ksrc = p.k + ld_koff + roff;
        const half_t* const vsrc = p.v + ld_voff + roff;
#pragma unroll
        for (int j = 0; j < (NDK + NPV + NW - 1) / NW; ++j) {
            const int pl = wave_u + NW * j;                      // wave-uniform
            if ((j + 1) * NW <= NDK + NPV || pl < NDK + NPV) {
                const bool isv = pl >= NDK;
                const int c = isv ? pl - NDK : pl;
                const half_t* src = rok ? (isv ? vsrc : ksrc) + c * 8 : zpage;
                const int dst = stage_half + (isv ? KAREA + c * VPL : c * KPL);
                glds16(src, __builtin_amdgcn_readfirstlane(smem_lds + 2u * (unsigned)dst));
            }
        }
        if (++ld_t == ld_ntile) {
            ld_t = 0;
            ++ld_s;
            if (ld_s < nsrc_eff) {
                const long off = (long)__builtin_amdgcn_readfirstlane(p.src_idx[bf * p.nsrc + ld_s]) * p.Nkv * p.ldkv + h * D;
                ld_koff = off; ld_voff = off;
            } else if (ld_s < nseg) {
                const long off = (long)__builtin_amdgcn_readfirstlane(p.x_idx[bf]) * p.Nkv_x * p.ldkv_x + h * D;
                ld_koff = off + xk_delta; ld_voff = off + xv_delta; ld_ld = p.ldkv_x; ld_nkv = p.Nkv_x; ld_ntile = ntile_x;
            }
        }
    };

    // ---- the pipeline pieces
    const int kf_off = g * KPL + l15 * 8;
    const int vf_off = KAREA + ((l15 & 3) >> 1) * VPL + (g * 4 + (l15 >> 2)) * 8 + (l15 & 1) * 4;
    auto kfrag_read = [&](const half_t* st, int hh, h8 (&kf)[2][KS]) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) kf[kb][ks] = *reinterpret_cast<const h8*>(&st[kf_off + (hh * 32 + kb * 16) * 8 + ks * 4 * KPL]);
    };
    auto vfrag_read = [&](const half_t* st, int hh, h8 (&vf)[DV16]) {
#pragma unroll
        for (int dv = 0; dv < DV16; ++dv) {
            const half_t* vp = &st[vf_off + hh * 32 * 8 + dv * 2 * VPL];
            fh4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fh4*)(vp));
            fh4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fh4*)(vp + 16 * 8));
            h8 a;
            a[0] = (half_t)lo[0]; a[1] = (half_t)lo[1]; a[2] = (half_t)lo[2]; a[3] = (half_t)lo[3];
            a[4] = (half_t)hi[0]; a[5] = (half_t)hi[1]; a[6] = (half_t)hi[2]; a[7] = (half_t)hi[3];
            vf[dv] = a;
        }
    };
    auto qk = [&](const h8 (&kf)[2][KS], f4 (&sc)[2][QB]) {       // S^T - M + lw of one 32-key step: 2 KS QB MFMAs, the first 2 QB start from cfold
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) sc[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kb][0], qf[qb][0], cfold[qb], 0, 0, 0);
#pragma unroll
        for (int ks = 1; ks < KS; ++ks)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) sc[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kb][ks], qf[qb][ks], sc[kb][qb], 0, 0, 0);
    };
    auto exp_part = [&](const f4 (&sc)[2][QB], h8 (&pb)[QB]) {    // P^T (fp16, B operand of the PV MFMA)
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            union { fh2 h[4]; h8 v; } u;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const float e0 = __builtin_amdgcn_exp2f(sc[kb][qb][0]), e1 = __builtin_amdgcn_exp2f(sc[kb][qb][1]);
                const float e2 = __builtin_amdgcn_exp2f(sc[kb][qb][2]), e3 = __builtin_amdgcn_exp2f(sc[kb][qb][3]);
                u.h[kb * 2] = __builtin_amdgcn_cvt_pkrtz(e0, e1);
                u.h[kb * 2 + 1] = __builtin_amdgcn_cvt_pkrtz(e2, e3);
            }
            pb[qb] = u.v;
        }
    };
    auto row_sums = [&](const h8 (&pb)[QB]) {                       // denominators from the packed probabilities: v_dot2c_f32_f16 against (1, 1)
        const fh2 one2 = {(__fp16)1.f, (__fp16)1.f};
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            union { fh2 h[4]; h8 v; } u;
            u.v = pb[qb];
#pragma unroll
            for (int i = 0; i < 4; ++i) lsum[qb] = __builtin_amdgcn_fdot2(u.h[i], one2, lsum[qb], false);
        }
    };
    auto pv = [&](const h8 (&vf)[DV16], const h8 (&pb)[QB]) {      // O^T += V^T P^T: 16 MFMAs
#pragma unroll
        for (int dv = 0; dv < DV16; ++dv)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) o[dv][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[dv], pb[qb], o[dv][qb], 0, 0, 0);
    };
    auto local_max = [&](const f4 (&sc)[2][QB], float (&mx)[QB]) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            float m = max3f(sc[0][qb][0], sc[0][qb][1], sc[0][qb][2]);
            m = max3f(m, sc[0][qb][3], sc[1][qb][0]);
            m = max3f(m, sc[1][qb][1], sc[1][qb][2]);
            mx[qb] = max3f(m, sc[1][qb][3], sc[1][qb][3]);
        }
    };
    auto mask_tail = [&](f4 (&sc)[2][QB], int key0, int nkv) {     // keys >= nkv of a segment's tail tile -> -inf
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (key0 + kb * 16 + g * 4 + r >= nkv) sc[kb][qb][r] = -INFINITY;
    };
    // reference update for the scores `sc` of the NEXT step (deferred, T13; see attn_pp40_kernel::decide).  M is quantised to 1/64
    // (softmax is shift invariant); the pending scores move by the same delta, O^T and the row sums are rescaled once per change,
    // and cfold takes the new reference for every later step.
    auto decide = [&](f4 (&sc)[2][QB], const float (&mx)[QB], float lw, bool first) {
        if (!first) {
            const float mall = QB == 4 ? fmaxf(max3f(mx[0], mx[1], mx[2]), mx[QB - 1]) : fmaxf(mx[0], mx[1]);
            if (__builtin_amdgcn_ballot_w64(mall > DEFER) == 0) return;
        }
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            if (__builtin_amdgcn_ballot_w64(first || mx[qb] > DEFER) != 0) {
                float m = mx[qb];
                const float o16 = __shfl_xor(m, 16, 64);
                m = max3f(m, o16, o16);
                const float o32 = __shfl_xor(m, 32, 64);
                m = max3f(m, o32, o32);
                float delta = floorf(m * 64.f + 0.5f) * (1.f / 64.f);              // shifted row max, quantised
                if (!first) delta = fmaxf(delta, 0.f);
                if (!(delta > -3.0e38f)) delta = 0.f;                                // a fully masked first step (-inf): keep M
                mrun[qb] += delta;
                const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
                const float cnew = lw - mrun[qb];
                cfold[qb] = f4{cnew, cnew, cnew, cnew};
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    sc[kb][qb][0] -= delta; sc[kb][qb][1] -= delta; sc[kb][qb][2] -= delta; sc[kb][qb][3] -= delta;
                }
                lsum[qb] *= alpha;
#pragma unroll
                for (int dv = 0; dv < DV16; ++dv) {
                    o[dv][qb][0] *= alpha; o[dv][qb][1] *= alpha; o[dv][qb][2] *= alpha; o[dv][qb][3] *= alpha;
                }
            }
        }
    };
    auto set_lw = [&](float lw) {                                  // the next source has another multiplicity
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const float cnew = lw - mrun[qb];
            cfold[qb] = f4{cnew, cnew, cnew, cnew};
        }
    };
// issue order inside the two overlapped regions (LLVM SchedGroupMask: VALU 0x2, MFMA 0x8, DS read 0x100, TRANS 0x400)
/* phase 1: 2 KS QB QK^T MFMAs over 2 DV16 V fragment reads, 8 QB exponentials and 4 QB conversions; phase 2: DV16 QB PV MFMAs over 2 KS K fragment */ \
/* reads and 8 QB max3 / dot2 (D = 64, QB = 4: 16 | 8, 32, 16 and 16 | 4, 32 — the round-3 pattern) */ \
#define UV_P64_PHASE1()                                                           \
    _Pragma("unroll") for (int i_ = 0; i_ < 2 * KS * QB; ++i_) {                  \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        \
        if (i_ < 2 * DV16) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     \
        if (i_ < 4 * QB) {                                                        \
            __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);                    \
            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                    \
        }                                                                         \
    }
#define UV_P64_PHASE2()                                                           \
    _Pragma("unroll") for (int i_ = 0; i_ < DV16 * QB; ++i_) {                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        \
        if (i_ < 2 * KS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       \
        if (i_ < 4 * QB) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);       \
    }
#define UV_P64_PIN4(a) do { if constexpr (QB == 4) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])); else asm volatile("" : "+v"(a[0]), "+v"(a[1])); } while (0)

    if constexpr (NPK > NDK) {      // the padding planes of K (columns D .. 32 KS - 1) are zero for the whole launch: the DMAs never write them
        for (int i = tid * 8; i < NST * TILE; i += NW * 64 * 8) *reinterpret_cast<h8*>(&smem[i]) = zero8;
        __syncthreads();
    }
    // ---- prologue: tiles 0 and 1 into the ring, scores + reference of step (0, 0)
    dma_tile(0);
    if (T > 1) dma_tile(TILE);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f4 scA[2][QB], scB[2][QB];
    h8 kf[2][KS], vf[DV16], pb[QB];
    float mx[QB];
    int cs_s = 0, cs_t = 0;                                  // (segment, tile in segment) of tile tt
    int nkv_cur = seg_nkv(0), nkv_nxt = nkv_cur;
    int nx_s = 0, nx_t = 0;
    auto advance_nx = [&]() {
        if (++nx_t == seg_ntile(nx_s)) { nx_t = 0; ++nx_s; }
    };
    advance_nx();                                            // -> tile 1
    float lw_nxt = nx_s < nseg ? seg_lw(nx_s) : lw_cur;
    nkv_nxt = nx_s < nseg ? seg_nkv(nx_s) : nkv_cur;
    int t0_cur = 0, t0_nxt = nx_t * KT;
    half_t* b_cur = smem;
    half_t* b_nxt = smem + TILE;
    half_t* b_ld = smem + 2 * TILE;
    (void)cs_s; (void)cs_t;

    kfrag_read(b_cur, 0, kf);
    qk(kf, scA);
    if (t0_cur + KT > nkv_cur) mask_tail(scA, t0_cur, nkv_cur);
    local_max(scA, mx);
    decide(scA, mx, lw_cur, true);
    kfrag_read(b_cur, 1, kf);

    for (int tt = 0; tt < T; ++tt) {
        const bool has_next = tt + 1 < T;
        if (tt + 2 < T) dma_tile((int)(b_ld - smem));        // tile tt+2 straight into the stage tile tt-1 left at the last barrier
        // ---- step (tt, 0): scores of (tt, 1) on the matrix pipe while (tt, 0) is exponentiated
        vfrag_read(b_cur, 0, vf);
        qk(kf, scB);
        exp_part(scA, pb);
        UV_P64_PIN4(pb);
        UV_P64_PHASE1();
        __builtin_amdgcn_sched_barrier(0);
        if (t0_cur + KT > nkv_cur) mask_tail(scB, t0_cur + 32, nkv_cur);
        kfrag_read(b_nxt, 0, kf);
        pv(vf, pb);
        local_max(scB, mx);
        row_sums(pb);
        UV_P64_PIN4(mx);
        UV_P64_PIN4(lsum);
        UV_P64_PHASE2();
        __builtin_amdgcn_sched_barrier(0);
        decide(scB, mx, lw_cur, false);
        // ---- step (tt, 1): scores of (tt+1, 0)
        if (lw_nxt != lw_cur) set_lw(lw_nxt);
        vfrag_read(b_cur, 1, vf);
        qk(kf, scA);
        exp_part(scB, pb);
        UV_P64_PIN4(pb);
        UV_P64_PHASE1();
        __builtin_amdgcn_sched_barrier(0);
        if (has_next && t0_nxt + KT > nkv_nxt) mask_tail(scA, t0_nxt, nkv_nxt);
        kfrag_read(b_nxt, 1, kf);
        pv(vf, pb);
        local_max(scA, mx);
        row_sums(pb);
        UV_P64_PIN4(mx);
        UV_P64_PIN4(lsum);
        UV_P64_PHASE2();
        __builtin_amdgcn_sched_barrier(0);
        if (has_next) decide(scA, mx, lw_nxt, false);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of tile tt+2 have landed
        __syncthreads();
        half_t* tb = b_cur; b_cur = b_nxt; b_nxt = b_ld; b_ld = tb;
        t0_cur = t0_nxt;
        nkv_cur = nkv_nxt;
        lw_cur = lw_nxt;
        advance_nx();
        if (nx_s < nseg) {
            lw_nxt = seg_lw(nx_s);
            nkv_nxt = seg_nkv(nx_s);
        }
        t0_nxt = nx_t * KT;
    }
#undef UV_P64_PHASE1
#undef UV_P64_PHASE2
#undef UV_P64_PIN4

    // ---- finalize: O^T[d = dv*16 + g*4 + r][q = l15] / l
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        float l = lsum[qb];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        float inv = 1.f / l, w1 = 0.f;
        const int qrow = qblk * RB + wave * 16 * QB + qb * 16 + l15;
        if (qrow >= p.Nq) continue;
        half_t* op = p.o + ((long)bf * p.Nq + qrow) * p.ldo + h * D;
        if constexpr (TP != 0) {          // two-phase attention (see attn_body): the reference M is in log2 units here (prescaled q, reference in the accumulator)
            const long srow = (((long)bf * p.heads + h) * p.Nq + qrow) * 2;
            if constexpr (TP == 1) {
                if (g == 0) *reinterpret_cast<float2*>(p.state_out + srow) = make_float2(mrun[qb], l);
            } else {
                const float2 st = *reinterpret_cast<const float2*>(p.state_in + srow);
                attn_merge_coef(st.x, st.y, mrun[qb], l, w1, inv);
            }
        }
#pragma unroll
        for (int dv = 0; dv < DV16; ++dv) {
            h4 ov;
            if constexpr (TP == 2) {
                const h4 o1 = *reinterpret_cast<const h4*>(op + dv * 16 + g * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) ov[r] = (half_t)fmaf((float)o1[r], w1, o[dv][qb][r] * inv);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) ov[r] = (half_t)(o[dv][qb][r] * inv);
            }
            *reinterpret_cast<h4*>(op + dv * 16 + g * 4) = ov;
        }
    }
}

// lane l of a 64-lane wave reports what ds_read_b64_tr_b16 returned for a known LDS image (bring-up aid).
__global__ void tr16_probe_kernel(float* out) {
    __shared__ __attribute__((aligned(16))) half_t sm[64 * 16];
    for (int i = threadIdx.x; i < 64 * 16; i += 64) sm[i] = (half_t)(float)i;   // value = row*16 + col
    __syncthreads();
    const int lane = threadIdx.x, l15 = lane & 15, g = lane >> 4;
    const half_t* vp = &sm[(g * 4 + (l15 >> 2)) * 16 + (l15 & 3) * 4];
    fh4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fh4*)(vp));
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (float)v[j];
}

// ----------------------------------------------------------------------------------------------------------------
// Text cross-attention (attn2: one source of <= 80 keys, head_dim 40 / 80 — 10 of the 16 launches of a UNet call, 0.9 ms of
// the step in the generic kernels, which pad 77 keys to two 64-key tiles, ring them through LDS with a barrier per tile and
// give every block a zero-fill prologue for 256 queries of work).  Here K and V of the (branch, head) are staged ONCE per block
// and then live in REGISTERS as MFMA A operands (5 key fragments x KS k-steps of K, DV16 x (32 + 32 + 16 keys) of V^T through
// ds_read_b64_tr_b16); a wave then streams 32 query rows per iteration with no further barrier or LDS traffic: Q fragments
// straight from memory (the B operand of S^T = K Q^T), all 80 scores of a row in registers (exact row max, no online
// rescaling), P^T packed in place as the B operand of O^T = V^T P^T (80 keys = three 32-key steps, the last half empty) and 8-byte
// stores of O.  What is left is the read of Q and the write of O.
template <int D>
__global__ __launch_bounds__(256, 2) void attn_text_kernel(AttnParams p) {
    constexpr int KS = (D + 31) / 32, DV16 = (D + 15) / 16, NKF = 5, NKEY = NKF * 16, NIT = 8;
    constexpr int KSTR = KS * 32 + 16;                       // halfs: 160 B (d=40) / 224 B (d=80) rows, conflict-free ds_read_b128
    constexpr int VSTR = DV16 * 16;                          // 96 B / 160 B rows, conflict-free ds_read_b64_tr_b16
    constexpr int DCH = D / 8;
    __shared__ __attribute__((aligned(16))) half_t smem[NKEY * KSTR + NKEY * VSTR];
    half_t* const Ks = smem;
    half_t* const Vs = smem + NKEY * KSTR;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int nchunk = (p.Nq + 128 * NIT - 1) / (128 * NIT);
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int chunk = lid % nchunk;
    const int h = (lid / nchunk) % p.heads;
    const int bf = lid / (nchunk * p.heads);
    const float c = p.q_prescaled ? 1.f : p.scale_log2e;
    const h8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    // ---- K / V of this (source block, head) -> LDS (zero padded to 80 keys and to the fragment widths), once
    for (int i = tid * 8; i < NKEY * KSTR + NKEY * VSTR; i += 256 * 8) *reinterpret_cast<h8*>(&smem[i]) = zero8;
    __syncthreads();
    {
        const long kvrow0 = (long)p.src_idx[bf * p.nsrc] * p.Nkv;
        for (int idx = tid; idx < p.Nkv * DCH; idx += 256) {
            const int row = idx / DCH, ch = idx - row * DCH;
            const long off = (kvrow0 + row) * p.ldkv + h * D + ch * 8;
            *reinterpret_cast<h8*>(&Ks[row * KSTR + ch * 8]) = *reinterpret_cast<const h8*>(p.k + off);
            *reinterpret_cast<h8*>(&Vs[row * VSTR + ch * 8]) = *reinterpret_cast<const h8*>(p.v + off);
        }
    }
    __syncthreads();
    h8 kf[NKF][KS];
#pragma unroll
    for (int kfi = 0; kfi < NKF; ++kfi)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kf[kfi][ks] = *reinterpret_cast<const h8*>(&Ks[(kfi * 16 + l15) * KSTR + ks * 32 + g * 8]);
    // V^T fragments: lane (d = dv*16 + l15, g) gets keys base + g*4 + {0..3} (lo) and base + 16 + g*4 + {0..3} (hi): the same key
    // order in which a lane's score registers sit, so P^T goes into the PV MFMA without leaving the lane
    h8 vf[DV16][2];
    h8 vf4[DV16];
    const int vf_off = (g * 4 + (l15 >> 2)) * VSTR + (l15 & 3) * 4;
#pragma unroll
    for (int dv = 0; dv < DV16; ++dv) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const half_t* vp = &Vs[vf_off + t * 32 * VSTR + dv * 16];
            fh4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fh4*)(vp));
            fh4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fh4*)(vp + 16 * VSTR));
            h8 a;
            a[0] = (half_t)lo[0]; a[1] = (half_t)lo[1]; a[2] = (half_t)lo[2]; a[3] = (half_t)lo[3];
            a[4] = (half_t)hi[0]; a[5] = (half_t)hi[1]; a[6] = (half_t)hi[2]; a[7] = (half_t)hi[3];
            vf[dv][t] = a;
        }
        // keys 64..79: the upper half of a third 32-key step is zero on both operands.  (The 16x16x16 MFMA would do these 16 keys
        // in half the matrix time, but chained behind v_mfma_f32_16x16x32_f16 on the same accumulator it returned stale values in
        // two of the four result registers on this toolchain — a missing wait state between the gfx950 and the legacy opcode.)
        fh4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fh4*)(&Vs[vf_off + 64 * VSTR + dv * 16]));
        h8 b = zero8;
        b[0] = (half_t)lo[0]; b[1] = (half_t)lo[1]; b[2] = (half_t)lo[2]; b[3] = (half_t)lo[3];
        vf4[dv] = b;
    }

    // ---- 32 query rows per iteration and wave
    const f4 z = {0.f, 0.f, 0.f, 0.f};
    auto load_q = [&](int q0, h8 (&qf)[2][KS]) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int qrow = q0 + qb * 16 + l15;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int dc = ks * 32 + g * 8;
                qf[qb][ks] = (qrow < p.Nq && dc < D) ? *reinterpret_cast<const h8*>(p.q + ((long)bf * p.Nq + qrow) * p.ldq + h * D + dc) : zero8;
            }
        }
    };
    h8 qf[2][KS], qn[2][KS];
    load_q(chunk * NIT * 128 + wave * 32, qf);
    for (int it = 0; it < NIT; ++it) {
        const int q0 = (chunk * NIT + it) * 128 + wave * 32;
        if (q0 >= p.Nq) break;
        // the next iteration's Q fragments are requested before this iteration's stores: loads issued after a store would wait for
        // its write acknowledgement (one in-order vmcnt), and nothing else hides the Q latency inside a wave
        if (it + 1 < NIT) load_q(q0 + 128, qn);
        f4 sc[NKF][2];
#pragma unroll
        for (int kfi = 0; kfi < NKF; ++kfi)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                f4 a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kfi][0], qf[qb][0], z, 0, 0, 0);
#pragma unroll
                for (int ks = 1; ks < KS; ++ks) a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kfi][ks], qf[qb][ks], a, 0, 0, 0);
                sc[kfi][qb] = a;
            }
#pragma unroll
        for (int kfi = 0; kfi < NKF; ++kfi)                    // keys >= Nkv -> -inf (kfi*16 + 15 < Nkv for all but the last fragments)
            if (kfi * 16 + 16 > p.Nkv) {
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kfi * 16 + g * 4 + r >= p.Nkv) sc[kfi][qb][r] = -INFINITY;
            }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float m = max3f(sc[0][qb][0], sc[0][qb][1], sc[0][qb][2]);
            m = max3f(m, sc[0][qb][3], sc[1][qb][0]);
#pragma unroll
            for (int kfi = 1; kfi < NKF; ++kfi) {
                m = max3f(m, sc[kfi][qb][1], sc[kfi][qb][2]);
                m = max3f(m, sc[kfi][qb][3], kfi + 1 < NKF ? sc[kfi + 1][qb][0] : sc[kfi][qb][3]);
            }
            const float o16 = __shfl_xor(m, 16, 64);
            m = max3f(m, o16, o16);
            const float o32 = __shfl_xor(m, 32, 64);
            m = max3f(m, o32, o32);
            const float mc = -m * c;
            float l = 0.f;
            union { fh2 h2v[4]; h8 v; } u0, u1;
            union { fh2 h2v[4]; h8 v; } u2;
            u2.v = zero8;
#pragma unroll
            for (int kfi = 0; kfi < NKF; ++kfi) {
                const float e0 = __builtin_amdgcn_exp2f(fmaf(sc[kfi][qb][0], c, mc)), e1 = __builtin_amdgcn_exp2f(fmaf(sc[kfi][qb][1], c, mc));
                const float e2 = __builtin_amdgcn_exp2f(fmaf(sc[kfi][qb][2], c, mc)), e3 = __builtin_amdgcn_exp2f(fmaf(sc[kfi][qb][3], c, mc));
                const fh2 p01 = __builtin_amdgcn_cvt_pkrtz(e0, e1), p23 = __builtin_amdgcn_cvt_pkrtz(e2, e3);
                // the denominator sums the SAME fp16-rounded weights that multiply V
                l += ((float)p01[0] + (float)p01[1]) + ((float)p23[0] + (float)p23[1]);
                if (kfi == 0) { u0.h2v[0] = p01; u0.h2v[1] = p23; }
                else if (kfi == 1) { u0.h2v[2] = p01; u0.h2v[3] = p23; }
                else if (kfi == 2) { u1.h2v[0] = p01; u1.h2v[1] = p23; }
                else if (kfi == 3) { u1.h2v[2] = p01; u1.h2v[3] = p23; }
                else { u2.h2v[0] = p01; u2.h2v[1] = p23; }
            }
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
            const float inv = 1.f / l;
            const int qrow = q0 + qb * 16 + l15;
            half_t* op = p.o + ((long)bf * p.Nq + qrow) * p.ldo + h * D;
#pragma unroll
            for (int dv = 0; dv < DV16; ++dv) {
                f4 o = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[dv][0], u0.v, z, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[dv][1], u1.v, o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf4[dv], u2.v, o, 0, 0, 0);
                const int dc = dv * 16 + g * 4;
                if (qrow < p.Nq && dc < D) {
                    h4 ov;
#pragma unroll
                    for (int r = 0; r < 4; ++r) ov[r] = (half_t)(o[r] * inv);
                    *reinterpret_cast<h4*>(op + dc) = ov;
                }
            }
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) qf[qb][ks] = qn[qb][ks];
    }
}

template <int DPAD, int DV16, int QB>
__global__ __launch_bounds__(256, 2) void attn_kernel_occ2(AttnParams p) {
    attn_body<DPAD, DV16, QB>(p);
}

// two-phase launches (AttnParams::state_out / state_in): their own instantiations (TP), so the epilogue of the one-launch kernels is untouched —
// the pipelined head_dim-40 kernel for long sequences with prescaled q, the generic body otherwise
template <int DPAD, int DV16, int TP>
int launch_attn_tp(const AttnParams& p, hipStream_t stream) {
    if constexpr (DPAD == 64 && DV16 == 3) {
        if (p.q_prescaled && p.Nq >= 2048) {
            const int nqb4 = (p.Nq + 255) / 256;
            hipLaunchKernelGGL((attn_pp40_kernel<true, 0, 1, true, true, TP>), dim3(nqb4 * p.heads * p.BF), dim3(256), 0, stream, p);
            UV_LAUNCH_CHECK();
            return UV_OK;
        }
    }
    if constexpr (DPAD == 64 && DV16 == 4) {      // head_dim 64 (the SD3 joint attention of ranks > 0): the pipelined kernel, as launch_attn picks it
        static const int pp64 = getenv("UNIVST_ATTN_PP64") ? atoi(getenv("UNIVST_ATTN_PP64")) : 2;
        // (the text queries of a joint attention — 333 rows — take it too, in the merge phase even without the text-key segment)
        if (pp64 && p.q_prescaled && (p.Nq >= 1024 || (pp64 == 2 && (p.kx || TP == 2) && p.Nq >= 192))) {
            const int nqb4 = (p.Nq + 255) / 256;
            hipLaunchKernelGGL((attn_pp64_kernel<64, 4, 0, TP>), dim3(nqb4 * p.heads * p.BF), dim3(256), 0, stream, p);
            UV_LAUNCH_CHECK();
            return UV_OK;
        }
    }
    const int QB = p.Nq >= 512 ? 2 : 1;
    const int nqb = (p.Nq + 64 * QB - 1) / (64 * QB);
    dim3 grid(nqb * p.heads * p.BF), block(256);
    if constexpr (DPAD >= 96) {
        if (p.q_prescaled) {
            if (QB == 2) {
                if constexpr (DPAD == 96) hipLaunchKernelGGL((attn_kernel_occ3<DPAD, DV16, 2, true, TP>), grid, block, 0, stream, p);
                else hipLaunchKernelGGL((attn_kernel<DPAD, DV16, 2, true, TP>), grid, block, 0, stream, p);
            } else {
                hipLaunchKernelGGL((attn_kernel<DPAD, DV16, 1, true, TP>), grid, block, 0, stream, p);
            }
            UV_LAUNCH_CHECK();
            return UV_OK;
        }
    }
    if (QB == 2) {
        if constexpr (DPAD <= 96) hipLaunchKernelGGL((attn_kernel_occ3<DPAD, DV16, 2, false, TP>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((attn_kernel<DPAD, DV16, 2, false, TP>), grid, block, 0, stream, p);
    } else {
        hipLaunchKernelGGL((attn_kernel<DPAD, DV16, 1, false, TP>), grid, block, 0, stream, p);
    }
    UV_LAUNCH_CHECK();
    return UV_OK;
}

template <int DPAD, int DV16>
int launch_attn(const AttnParams& p, hipStream_t stream) {
    if (p.state_out) return launch_attn_tp<DPAD, DV16, 1>(p, stream);
    if (p.state_in) return launch_attn_tp<DPAD, DV16, 2>(p, stream);
    static const int qb4 = getenv("UNIVST_ATTN_QB4") ? atoi(getenv("UNIVST_ATTN_QB4")) : 1;   // 64 query rows per wave for long sequences
    if constexpr (DPAD == 64 && DV16 == 3) {
        // UNIVST_ATTN_PP (A/B aid): 2 = software-pipelined kernel with scale/max folded into the MFMA when q is prescaled
        // (default; plain q takes attn_body), 1 = software-pipelined with plain softmax arithmetic, 0 = attn_body
        static const int pp = getenv("UNIVST_ATTN_PP") ? atoi(getenv("UNIVST_ATTN_PP")) : 2;
        if (!p.kx && p.Nq >= 2048 && ((pp == 2 && p.q_prescaled) || pp == 1)) {
            const int nqb4 = (p.Nq + 255) / 256;
            const bool text = p.nsrc == 1 && p.Nkv <= 128;
            // UNIVST_ATTN_STG (A/B aid): 1 (default) = K/V ring filled by LDS-DMA + one wave-wide reference test + 16-wide second k step;
            // 2 = the same with four per-block tests, 3 = with the 32-wide second k step; 0 = the round-2 kernel (ring through registers)
            static const int stg = getenv("UNIVST_ATTN_STG") ? atoi(getenv("UNIVST_ATTN_STG")) : 1;
            if (pp == 2 && text && !p.state_out && !p.state_in) hipLaunchKernelGGL((attn_pp40_kernel<true, 1>), dim3(nqb4 * p.heads * p.BF), dim3(256), 0, stream, p);
            else if (pp == 2 && stg == 2) hipLaunchKernelGGL((attn_pp40_kernel<true, 0, 1, false, true>), dim3(nqb4 * p.heads * p.BF), dim3(256), 0, stream, p);
            else if (pp == 2 && stg == 3) hipLaunchKernelGGL((attn_pp40_kernel<true, 0, 1, true, false>), dim3(nqb4 * p.heads * p.BF), dim3(256), 0, stream, p);
            else if (pp == 2 && stg) hipLaunchKernelGGL((attn_pp40_kernel<true, 0, 1, true, true>), dim3(nqb4 * p.heads * p.BF), dim3(256), 0, stream, p);
            else if (pp == 2) hipLaunchKernelGGL((attn_pp40_kernel<true, 0>), dim3(nqb4 * p.heads * p.BF), dim3(256), 0, stream, p);
            else hipLaunchKernelGGL((attn_pp40_kernel<false, 0>), dim3(nqb4 * p.heads * p.BF), dim3(256), 0, stream, p);
            UV_LAUNCH_CHECK();
            return UV_OK;
        }
    }
    if constexpr (DPAD == 64 && DV16 == 4) {
        // head_dim 64 with prescaled q (the SD3 joint attention folds the factor into the q RMSNorm weight): software-pipelined kernel.
        // UNIVST_ATTN_PP64=0 (A/B aid): the generic body
        static const int pp64 = getenv("UNIVST_ATTN_PP64") ? atoi(getenv("UNIVST_ATTN_PP64")) : 2;
        // (pp64 = 2, default since round 4: also the text queries of a joint attention — 333 rows over 12 621 keys — which otherwise take the
        // generic body: SD3.5 step 944 / 947 -> 934 / 943 ms, same box; 1 = image queries only)
        if (pp64 && !p.state_out && !p.state_in && p.q_prescaled && (p.Nq >= 1024 || (pp64 == 2 && p.kx && p.Nq >= 192))) {
            const int nqb4 = (p.Nq + 255) / 256;
            hipLaunchKernelGGL((attn_pp64_kernel<64, 4, 0>), dim3(nqb4 * p.heads * p.BF), dim3(256), 0, stream, p);
            UV_LAUNCH_CHECK();
            return UV_OK;
        }
    }
    if constexpr (DPAD == 96 && DV16 == 5) {
        // head_dim 80 (the 32x32 level of SD-v1.5) with prescaled q: the same pipeline at 32 query rows per wave (round 5) — measured and NOT
        // dispatched by default.  It runs two waves per SIMD where attn_kernel_occ3 runs three, and with 44 MFMAs per wave between two tile
        // barriers the third wave hides more than the leaner instruction stream saves: same box, alternating, the 32x32-level attention of the
        // full step 2.18 -> 2.27 ms; of a rank's shard (384 blocks) 0.435 -> 0.404 ms — and the generic body with the same arithmetic savings
        // (attn_body CF, built right after) takes 1.86 ms / 0.364 ms.  UNIVST_ATTN_PP80 = 1: grids of at most one round, 2: always (A/B aid)
        static const int pp80 = getenv("UNIVST_ATTN_PP80") ? atoi(getenv("UNIVST_ATTN_PP80")) : 0;
        const int nqb2 = (p.Nq + 127) / 128;
        if (pp80 && !p.state_out && !p.state_in && p.q_prescaled && !p.kx && p.Nq >= 512 && (pp80 == 2 || (long)nqb2 * p.heads * p.BF <= 512)) {
            hipLaunchKernelGGL((attn_pp64_kernel<80, 2, 0>), dim3(nqb2 * p.heads * p.BF), dim3(256), 0, stream, p);
            UV_LAUNCH_CHECK();
            return UV_OK;
        }
    }
    // (probe, round 5) UNIVST_ATTN_CF40=1 with UNIVST_ATTN_PP=0: head_dim 40 on the generic body with the accumulator-folded reference at three waves per SIMD
    static const int cf40 = getenv("UNIVST_ATTN_CF40") ? atoi(getenv("UNIVST_ATTN_CF40")) : 0;
    if constexpr (DPAD == 64 && DV16 == 4) {      // (probe) UNIVST_ATTN_CF64=1 with UNIVST_ATTN_PP64=0: the same at head_dim 64 (SD3 joint attention, SD-v2.1)
        static const int cf64 = getenv("UNIVST_ATTN_CF64") ? atoi(getenv("UNIVST_ATTN_CF64")) : 0;
        if (cf64 && p.q_prescaled && p.Nq >= 192) {
            const int nqb2 = (p.Nq + 127) / 128;
            hipLaunchKernelGGL((attn_kernel_occ3<DPAD, DV16, 2, true>), dim3(nqb2 * p.heads * p.BF), dim3(256), 0, stream, p);
            UV_LAUNCH_CHECK();
            return UV_OK;
        }
    }
    if constexpr (DPAD == 64 && DV16 == 3) {
        if (cf40 && p.q_prescaled && p.Nq >= 512) {
            const int nqb2 = (p.Nq + 127) / 128;
            hipLaunchKernelGGL((attn_kernel_occ3<DPAD, DV16, 2, true>), dim3(nqb2 * p.heads * p.BF), dim3(256), 0, stream, p);
            UV_LAUNCH_CHECK();
            return UV_OK;
        }
    }
    if constexpr (DPAD <= 64) {
        if (qb4 && p.Nq >= 2048) {
            const int nqb4 = (p.Nq + 255) / 256;
            hipLaunchKernelGGL((attn_kernel_occ2<DPAD, DV16, 4>), dim3(nqb4 * p.heads * p.BF), dim3(256), 0, stream, p);
            UV_LAUNCH_CHECK();
            return UV_OK;
        }
    }
    const int QB = p.Nq >= 512 ? 2 : 1;
    const int nqb = (p.Nq + 64 * QB - 1) / (64 * QB);
    dim3 grid(nqb * p.heads * p.BF), block(256);
    if constexpr (DPAD >= 96) {
        // prescaled q on the wide heads (80, 160: the 32x32 / 16x16 levels of SD-v1.5): the reference rides in the MFMA accumulator, row sums by
        // v_dot2 (attn_body CF).  UNIVST_ATTN_CF=0 (A/B aid): the plain softmax arithmetic
        static const int cfenv = getenv("UNIVST_ATTN_CF") ? atoi(getenv("UNIVST_ATTN_CF")) : 1;
        if (cfenv && p.q_prescaled) {
            if constexpr (DPAD == 96) {
                if (QB == 2) hipLaunchKernelGGL((attn_kernel_occ3<DPAD, DV16, 2, true>), grid, block, 0, stream, p);
                else hipLaunchKernelGGL((attn_kernel<DPAD, DV16, 1, true>), grid, block, 0, stream, p);
            } else {
                if (QB == 2) hipLaunchKernelGGL((attn_kernel<DPAD, DV16, 2, true>), grid, block, 0, stream, p);
                else hipLaunchKernelGGL((attn_kernel<DPAD, DV16, 1, true>), grid, block, 0, stream, p);
            }
            UV_LAUNCH_CHECK();
            return UV_OK;
        }
    }
    if constexpr (DPAD <= 96) {           // (not instantiated for wider heads: at 168 VGPRs they spill, and a spilled prefetch
        if (QB == 2) {                    //  register is read before its asm load has landed — tests/test_asm_hazards.py)
            hipLaunchKernelGGL((attn_kernel_occ3<DPAD, DV16, 2>), grid, block, 0, stream, p);
            UV_LAUNCH_CHECK();
            return UV_OK;
        }
    }
    if (QB == 2) hipLaunchKernelGGL((attn_kernel<DPAD, DV16, 2>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((attn_kernel<DPAD, DV16, 1>), grid, block, 0, stream, p);
    UV_LAUNCH_CHECK();
    return UV_OK;
}

}  // namespace

static int attn_dispatch(const AttnParams& p, hipStream_t stream);

int uv_launch_attention(const AttnParams& p0, hipStream_t stream) {
    static const int order_env = getenv("UNIVST_ATTN_ORDER") ? atoi(getenv("UNIVST_ATTN_ORDER")) : 1;
    AttnParams p = p0;
    p.order = order_env;
    UV_REQUIRE(p.d % 8 == 0, "attention: head_dim=%d must be a multiple of 8", p.d);
    UV_REQUIRE(p.nsrc >= 1 && p.Nkv >= 1 && p.Nq >= 1, "attention: empty problem");
    UV_REQUIRE(p.ldq % 8 == 0 && p.ldkv % 8 == 0 && p.ldo % 4 == 0, "attention: row strides must be multiples of 8");
    UV_REQUIRE(!p.kx || (p.vx && p.x_idx && p.Nkv_x >= 1 && p.ldkv_x % 8 == 0), "attention: extra key segment needs vx, x_idx, Nkv_x >= 1 and a row stride that is a multiple of 8");
    UV_REQUIRE(!(p.state_out && p.state_in), "attention: a launch is the first phase (state_out) or the merge phase (state_in) of a two-phase attention, not both");
    const double nkv = (double)p.nsrc * p.Nkv;
    const int cls = (p.nsrc == 1 && p.Nkv <= 128) ? UV_CLS_ATTN_TEXT
                    : (p.d == 40 && p.Nq >= 2048) ? UV_CLS_ATTN_D40 : (p.d == 80 ? UV_CLS_ATTN_D80 : UV_CLS_ATTN_OTHER);
    // (a two-phase attention is ONE algorithmic attention over the reference's key set: the first phase is charged its flops, the merge phase none —
    // the host does not know how the sources split between the two)
    uv_prof_begin(cls, p.state_in ? 0.0 : 4.0 * p.BF * p.heads * (double)p.Nq * nkv * p.d,
                  2.0 * p.BF * p.heads * p.d * (2.0 * p.Nq + (p.state_in ? 0.0 : 2.0 * nkv)), stream);
    int rc = attn_dispatch(p, stream);
    uv_prof_end(stream);
    return rc;
}

static int attn_dispatch(const AttnParams& p, hipStream_t stream) {
    // text cross-attention: one short source, K/V held in registers (attn_text_kernel).  UNIVST_ATTN_TEXT=0: generic kernels (A/B aid)
    static const int text_env = getenv("UNIVST_ATTN_TEXT") ? atoi(getenv("UNIVST_ATTN_TEXT")) : 1;
    if (text_env && !p.kx && !p.state_out && !p.state_in && p.nsrc == 1 && p.Nkv <= 80 && !p.src_logw && (p.d == 40 || p.d == 80) && p.Nq >= 256) {
        const int nchunk = (p.Nq + 1023) / 1024;
        const dim3 grid((unsigned)(nchunk * p.heads * p.BF));
        if (p.d == 40) hipLaunchKernelGGL((attn_text_kernel<40>), grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((attn_text_kernel<80>), grid, dim3(256), 0, stream, p);
        UV_LAUNCH_CHECK();
        return UV_OK;
    }
    switch (p.d) {

        case 16: return launch_attn<32, 1>(p, stream);
        case 32: return launch_attn<32, 2>(p, stream);
        case 40: return launch_attn<64, 3>(p, stream);
        case 64: return launch_attn<64, 4>(p, stream);
        case 80: return launch_attn<96, 5>(p, stream);
        case 160: return launch_attn<160, 10>(p, stream);
        default:
            uv_set_error("attention: head_dim=%d not instantiated (16,32,40,64,80,160)", p.d);
            return UV_ERR_UNSUPPORTED;
    }
}

int uv_launch_tr16_probe(float* out, hipStream_t stream) {
    hipLaunchKernelGGL(tr16_probe_kernel, dim3(1), dim3(64), 0, stream, out);
    UV_LAUNCH_CHECK();
    return UV_OK;
}
