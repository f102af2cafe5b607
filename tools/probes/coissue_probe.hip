// Micro-probe (not part of the product): what do the two waves of a SIMD share?  A 512-thread workgroup (waves w and w+4 share a
// SIMD) runs, per iteration and wave, the head_dim-40 attention step's instruction multiset: 28 x v_mfma_f32_16x16x32_f16 and the
// softmax VALU work (32 v_exp_f32, 16 v_cvt_pkrtz, 16 v_max3).
//   mode 0: waves 0..3 only, MFMA stream only                     -> matrix-pipe time of a step
//   mode 1: waves 4..7 only, VALU stream only                     -> VALU time of a step
//   mode 2: waves 0..3 MFMA stream, waves 4..7 VALU stream, free-running   -> do the two pipes overlap across waves of a SIMD?
//   mode 3: as 2 with a workgroup barrier per iteration (the ping-pong schedule)
//   mode 4: all 8 waves, each MFMA block then VALU block per iteration (roles NOT split; natural de-phasing)
//   mode 5: all 8 waves, fine interleave 1 MFMA : ~2.3 VALU in program order
//   mode 6: waves 0..3 only, interleaved stream (one wave per SIMD)
//   modes 7 / 8: mode 3 with the roles assigned by wave bit 0 / bit 1 instead of bit 2 (checks which waves share a SIMD)
// Reported: ns per (wave-step, SIMD) = time / (iterations x steps per SIMD-iteration) so that all modes are comparable:
// modes 0/1/6 execute ONE wave-step per SIMD and iteration, modes 2/3 one MFMA half + one VALU half (= one step), modes 4/5 two.
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __fp16 fh2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float max3f(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

struct State {
    h8 a[4], b[4], pb[4];
    f4 acc[7][4];
    f4 sc[2][4];
    float m[4];
};

__device__ __forceinline__ void mfma_block(State& s) {
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(s.a[j]));
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s.acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(s.a[i & 3], s.pb[j], s.acc[i][j], 0, 0, 0);
}
__device__ __forceinline__ void valu_block(State& s) {
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) s.m[qb] = max3f(s.sc[0][qb][0], s.sc[0][qb][1], s.sc[0][qb][2]);
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) s.m[qb] = max3f(s.m[qb], s.sc[0][qb][3], s.sc[1][qb][0]);
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) s.m[qb] = max3f(s.m[qb], s.sc[1][qb][1], s.sc[1][qb][2]);
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) s.m[qb] = max3f(s.m[qb], s.sc[1][qb][3], s.sc[1][qb][3]);
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) {
        union { fh2 h[4]; h8 v; } u;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const float e0 = __builtin_amdgcn_exp2f(s.sc[kb][qb][0]), e1 = __builtin_amdgcn_exp2f(s.sc[kb][qb][1]);
            const float e2 = __builtin_amdgcn_exp2f(s.sc[kb][qb][2]), e3 = __builtin_amdgcn_exp2f(s.sc[kb][qb][3]);
            u.h[kb * 2] = __builtin_amdgcn_cvt_pkrtz(e0, e1);
            u.h[kb * 2 + 1] = __builtin_amdgcn_cvt_pkrtz(e2, e3);
        }
        s.pb[qb] = u.v;
        asm volatile("" : "+v"(s.pb[qb]), "+v"(s.m[qb]));
        // feed the scores back so that the loop carries a dependence (nothing is hoisted) without extra VALU work: the
        // next iteration exponentiates the running max chain's inputs again
        asm volatile("" : "+v"(s.sc[0][qb]), "+v"(s.sc[1][qb]));
    }
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void probe(float* out, const _Float16* in, int iters) {
    State s;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), grp = wave >> 2;
    for (int i = 0; i < 4; ++i) {
        s.a[i] = *reinterpret_cast<const h8*>(in + (tid + i * 512) * 8);
        s.b[i] = *reinterpret_cast<const h8*>(in + (tid + (4 + i) * 512) * 8);
        s.pb[i] = s.b[i];
    }
    for (int i = 0; i < 7; ++i) for (int j = 0; j < 4; ++j) s.acc[i][j] = f4{0, 0, 0, 0};
    for (int j = 0; j < 4; ++j) s.m[j] = 0.f;
    for (int k = 0; k < 2; ++k) for (int j = 0; j < 4; ++j)
        s.sc[k][j] = f4{(float)s.a[j][k * 4 + 0], (float)s.a[j][k * 4 + 1], (float)s.b[j][k * 4 + 2], (float)s.b[j][k * 4 + 3]};
    for (int k = 0; k < 2; ++k) for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(s.sc[k][j]));
    if (MODE == 0 && grp == 1) return;
    if (MODE == 1 && grp == 0) return;
    if (MODE == 6 && grp == 1) return;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            mfma_block(s);
        } else if (MODE == 1) {
            valu_block(s);
        } else if (MODE == 2 || MODE == 3 || MODE == 7 || MODE == 8) {
            const int role = MODE == 7 ? (wave & 1) : MODE == 8 ? ((wave >> 1) & 1) : grp;
            if (role == 0) mfma_block(s); else valu_block(s);
            if (MODE != 2) { __builtin_amdgcn_sched_barrier(0); __syncthreads(); __builtin_amdgcn_sched_barrier(0); }
        } else if (MODE == 4) {
            mfma_block(s);
            __builtin_amdgcn_sched_barrier(0);
            valu_block(s);
            __builtin_amdgcn_sched_barrier(0);
        } else {       // 5, 6: interleaved in program order
            mfma_block(s);
            valu_block(s);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            }
#pragma unroll
            for (int i = 0; i < 24; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float r = 0.f;
    for (int i = 0; i < 7; ++i) for (int j = 0; j < 4; ++j) r += s.acc[i][j][0] + s.acc[i][j][1] + s.acc[i][j][2] + s.acc[i][j][3];
    for (int j = 0; j < 4; ++j) r += (float)s.pb[j][0] + s.m[j];
    out[blockIdx.x * 512 + tid] = r;
}

template <int MODE>
static void run(float* out, const _Float16* in, int blocks, int iters, const char* what, double steps_per_simd_iter) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(512), 0, 0, out, in, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double rounds = blocks / 256.0;     // one 512-thread block per CU at a time
    const double ns = ms * 1e6 / (rounds * iters * steps_per_simd_iter);
    printf("mode %d  %-72s %8.3f ms  %7.1f ns per wave-step and SIMD\n", MODE, what, ms, ns);
}

int main() {
    const int blocks = 256 * 4, iters = 4000;
    float* out;
    _Float16* in;
    hipMalloc(&out, blocks * 512 * 4);
    hipMalloc(&in, 512 * 8 * 8 * 2);
    _Float16* h = (_Float16*)malloc(512 * 8 * 8 * 2);
    for (int i = 0; i < 512 * 8 * 8; ++i) h[i] = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
    hipMemcpy(in, h, 512 * 8 * 8 * 2, hipMemcpyHostToDevice);
    run<0>(out, in, blocks, iters, "MFMA stream only, one wave per SIMD (28 MFMA / step)", 1);
    run<1>(out, in, blocks, iters, "VALU stream only, one wave per SIMD (32 exp + 16 cvt + 16 max3 / step)", 1);
    run<2>(out, in, blocks, iters, "wave A MFMA stream || wave B VALU stream, free-running", 1);
    run<3>(out, in, blocks, iters, "wave A MFMA stream || wave B VALU stream, barrier per step (ping-pong)", 1);
    run<4>(out, in, blocks, iters, "both waves: MFMA block then VALU block (roles not split)", 2);
    run<5>(out, in, blocks, iters, "both waves: fine interleave 1 MFMA : 2.3 VALU", 2);
    run<6>(out, in, blocks, iters, "one wave per SIMD: fine interleave", 1);
    run<7>(out, in, blocks, iters, "as mode 3, roles by wave parity (w & 1)", 1);
    run<8>(out, in, blocks, iters, "as mode 3, roles by (w >> 1) & 1", 1);
    return 0;
}
