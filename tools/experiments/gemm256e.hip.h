// gemm256e.hip.h — 256x256 MFMA GEMM, 4 phases per K step, staggered wave groups, a
// continuous LDS-DMA stream with counted waits (gfx950).
//
//   C[M,N] = epilogue( A[M,K] · W[N,K]ᵀ )      (contract and epilogue of gemm.hip.h)
//
// Tile and wave layout as gemm256.hip.h (8 waves as 2(M) x 4(N), a wave owns 128x64 =
// 4x2 MFMA tiles of 32x32).  The K loop is cut finer:
//
//   * LDS holds 8 half-tile slots of 16 KiB: {A rows 0-127, A rows 128-255, W rows 0-127,
//     W rows 128-255} x two K steps; a slot image is 128 rows x 128 B with the 16-byte
//     chunk swizzle (row>>1)&7 applied on the DMA source address.
//   * A K step is four phases, one quadrant of the wave's tile each (8 MFMAs from
//     registers): (i01,j0) (i01,j1) (i23,j1) (i23,j0).  A phase is
//         LOAD part: 2 LDS-DMA requests (one half-tile per phase, all 8 waves share it) +
//                    the ds_reads of the fragments the quadrant still needs (12/4/8/0)
//         s_barrier
//         MATH part: 8 MFMAs
//         s_barrier
//     and the two wave groups (wave>>2) run one barrier apart, so while one group is in its
//     MATH part the other one is in its LOAD part: every SIMD always has a wave with MFMAs.
//   * DMA schedule of K step t: phase 0 -> A0(t+1), 1 -> A1(t+1), 2 -> W0(t+2), 3 -> W1(t+2):
//     a W slot is dead after phase 1 (both groups copied their W fragments), an A slot after
//     phase 2, so the stream never waits for a whole stage to drain; one counted
//     s_waitcnt vmcnt(4) per K step (never 0 in steady state) at the end of phase 3's LOAD
//     part makes A(t+1) visible, W(t+1) landed a step earlier.
//
// The K reduction order per accumulator is kk = 0..3 inside each step, as in the other
// kernels, so results are bit-identical to gemm.hip.h / gemm256.hip.h.
// EXPERIMENT (not part of libzett_hip.so): built only by tools/gemm_bench.hip; measured
// results and why the product does not use it are in DESIGN.md §4.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm256.hip.h"

namespace zett {

constexpr int E_HALF_BYTES = 128 * GEMM_ROW_BYTES;      // 16 KiB
constexpr int E_LDS_BYTES = 8 * E_HALF_BYTES;           // 128 KiB

__device__ __forceinline__ void e_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// barrier first, LDS-read wait after it: the fragment reads issued before the barrier complete while
// the wave waits for the other group (safe where the slots being read are not refilled for >= 2 barriers)
__device__ __forceinline__ void e_barrier_then_wait() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void e_barrier_vm4() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void e_barrier_vm0() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

template <typename T, int PRIO = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm256e_tn_kernel(GemmArgs<T> g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BK = GEMM_ROW_BYTES / (int)sizeof(T);

    const int tiles_m = (g.M + G256_BM - 1) / G256_BM;
    const int tiles_n = (g.N + G256_BN - 1) / G256_BN;
    const int nwg = tiles_m * tiles_n;
    int wg = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
    }
    constexpr int GROUP_M = 8;
    const int group_size = GROUP_M * tiles_n;
    const int first_m = (wg / group_size) * GROUP_M;
    const int gm = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
    const int tm = first_m + (wg % group_size) % gm;
    const int tn = (wg % group_size) / gm;
    const int m0 = tm * G256_BM, n0 = tn * G256_BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5;

    // DMA sources: half-tile h in {A0, A1, W0, W1}; this wave moves rows wave*16 + j*8 + lane/8 of it
    const unsigned char* src[4][2];
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = wave * 16 + j * 8 + (lane >> 3);            // row inside the half-tile
            const int ch = (lane & 7) ^ ((row >> 1) & 7);
            if (h < 2) {
                int r = m0 + h * 128 + row; r = r < g.M ? r : g.M - 1;
                src[h][j] = (const unsigned char*)(g.A + (size_t)r * g.lda) + ch * 16;
            } else {
                int r = n0 + (h - 2) * 128 + row; r = r < g.N ? r : g.N - 1;
                src[h][j] = (const unsigned char*)(g.W + (size_t)r * g.ldw) + ch * 16;
            }
        }
    const int dma_row_off = wave * 16 * GEMM_ROW_BYTES;

    auto issue_half = [&](int kt, int h) {      // half-tile h of K step kt -> slot ((kt&1)*4 + h)
        unsigned char* dst = smem + ((kt & 1) * 4 + h) * E_HALF_BYTES + dma_row_off;
        const size_t koff = (size_t)kt * GEMM_ROW_BYTES;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src[h][0] + koff), (lds_ptr_t)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src[h][1] + koff), (lds_ptr_t)(dst + 8 * GEMM_ROW_BYTES), 16, 0, 0);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment offsets inside a buffer (4 half-tile slots): A rows of half wm, W rows of half wn>>1
    const int row_swz = (l31 >> 1) & 7;
    int a_off[4], w_off[2], swz[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a_off[i] = wm * E_HALF_BYTES + (i * 32 + l31) * GEMM_ROW_BYTES;
#pragma unroll
    for (int j = 0; j < 2; ++j) w_off[j] = (2 + (wn >> 1)) * E_HALF_BYTES + ((wn & 1) * 64 + j * 32 + l31) * GEMM_ROW_BYTES;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) swz[kk] = ((kk * 2 + hi) ^ row_swz) << 4;

    u32x4 fa[2][4];       // [i within the pair][kk]
    u32x4 fb0[4], fb1[4]; // [kk]

    const int nk = g.K / BK;

    // prologue: K step 0 completely, W halves of step 1
#pragma unroll
    for (int h = 0; h < 4; ++h) issue_half(0, h);
    if (nk > 1) { issue_half(1, 2); issue_half(1, 3); e_barrier_vm4(); } else { e_barrier_vm0(); }
    if (wm == 1) e_barrier();                      // stagger: group 1 runs one barrier behind

    for (int t = 0; t < nk; ++t) {
        const unsigned char* S = smem + (t & 1) * 4 * E_HALF_BYTES;
        // ---- phase 0: quadrant (i0,i1) x j0
        if (t + 1 < nk) issue_half(t + 1, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            fa[0][kk] = *(const u32x4*)(S + a_off[0] + swz[kk]);
            fa[1][kk] = *(const u32x4*)(S + a_off[1] + swz[kk]);
            fb0[kk] = *(const u32x4*)(S + w_off[0] + swz[kk]);
        }
        if (PRIO == 2) e_barrier_then_wait(); else e_barrier();
        if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { mfma_chunk<T>(fa[0][kk], fb0[kk], acc[0][0]); mfma_chunk<T>(fa[1][kk], fb0[kk], acc[1][0]); }
        if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
        e_barrier();
        // ---- phase 1: (i0,i1) x j1
        if (t + 1 < nk) issue_half(t + 1, 1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) fb1[kk] = *(const u32x4*)(S + w_off[1] + swz[kk]);
        e_barrier();
        if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { mfma_chunk<T>(fa[0][kk], fb1[kk], acc[0][1]); mfma_chunk<T>(fa[1][kk], fb1[kk], acc[1][1]); }
        if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
        e_barrier();
        // ---- phase 2: (i2,i3) x j1
        if (t + 2 < nk) issue_half(t + 2, 2);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            fa[0][kk] = *(const u32x4*)(S + a_off[2] + swz[kk]);
            fa[1][kk] = *(const u32x4*)(S + a_off[3] + swz[kk]);
        }
        if (PRIO == 2) e_barrier_then_wait(); else e_barrier();
        if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { mfma_chunk<T>(fa[0][kk], fb1[kk], acc[2][1]); mfma_chunk<T>(fa[1][kk], fb1[kk], acc[3][1]); }
        if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
        e_barrier();
        // ---- phase 3: (i2,i3) x j0; the halves of step t+1 must be visible after its LOAD-part barrier
        if (t + 2 < nk) { issue_half(t + 2, 3); e_barrier_vm4(); } else { e_barrier_vm0(); }
        if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { mfma_chunk<T>(fa[0][kk], fb0[kk], acc[2][0]); mfma_chunk<T>(fa[1][kk], fb0[kk], acc[3][0]); }
        if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
        e_barrier();
    }
    if (wm == 0) e_barrier();                      // group 0 waits for group 1's last MATH part

    // ---- epilogue (see gemm256.hip.h)
    float* region = (float*)(smem + wave * 16384);
    const GemmEpilogue<T>& e = g.epi;
    const int c4 = (lane & 15) * 4;
    const int gcol = n0 + wn * 64 + c4;
    const bool col_ok = gcol < g.N;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f), sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = bias4;
    if (col_ok) {
        if (e.bias) bias4 = *(const float4*)(e.bias + gcol);
        if (e.scale) sc4 = *(const float4*)(e.scale + gcol);
        if (e.shift) sh4 = *(const float4*)(e.shift + gcol);
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    region[(i2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * 64 + j * 32 + l31] = acc[2 * p + i2][j][r];
        for (int t = 0; t < 16; ++t) {
            const int lrow = t * 4 + (lane >> 4);
            const int grow = m0 + wm * 128 + p * 64 + lrow;
            float4 v = *(const float4*)(region + lrow * 64 + c4);
            if (grow >= g.M || !col_ok) continue;
            v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
            if (e.act == ACT_GELU_TANH) { v.x = gelu_tanh_f(v.x); v.y = gelu_tanh_f(v.y); v.z = gelu_tanh_f(v.z); v.w = gelu_tanh_f(v.w); }
            else if (e.act == ACT_GELU_ERF) { v.x = gelu_erf_f(v.x); v.y = gelu_erf_f(v.y); v.z = gelu_erf_f(v.z); v.w = gelu_erf_f(v.w); }
            if (e.residual) {
                const float4 rr = *(const float4*)(e.residual + (size_t)grow * e.ld_res + gcol);
                v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
            }
            if (e.scale) { v.x = sc4.x * v.x + sh4.x; v.y = sc4.y * v.y + sh4.y; v.z = sc4.z * v.z + sh4.z; v.w = sc4.w * v.w + sh4.w; }
            if (gcol < e.split_col) {
                if (e.out_f32) *(float4*)(e.out_f32 + (size_t)grow * e.ld_f32 + gcol) = v;
                if (e.out_lo) store_out4<T>(e.out_lo + (size_t)grow * e.ld_lo + gcol, v);
            } else if (e.out_f32_b) {
                *(float4*)(e.out_f32_b + (size_t)grow * e.ld_f32 + (gcol - e.split_col)) = v;
            }
        }
    }
}

template <typename T, int PRIO = 0>
inline hipError_t launch_gemm256e(const GemmArgs<T>& g, hipStream_t stream) {
    const int tiles_m = (g.M + G256_BM - 1) / G256_BM;
    const int tiles_n = (g.N + G256_BN - 1) / G256_BN;
    if (tiles_m <= 0 || tiles_n <= 0) return hipSuccess;
    hipLaunchKernelGGL((gemm256e_tn_kernel<T, PRIO>), dim3(tiles_m * tiles_n), dim3(512), E_LDS_BYTES, stream, g);
    return hipGetLastError();
}

}  // namespace zett
