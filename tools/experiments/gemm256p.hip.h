// gemm256p.hip.h — 256x256 MFMA GEMM with ping-pong wave groups (gfx950).
//
//   C[M,N] = epilogue( A[M,K] · W[N,K]ᵀ )      (contract and epilogue of gemm.hip.h)
//
// Same tile, LDS image and LDS-DMA staging as gemm256.hip.h (256x256 tile, 128 bytes of K
// per row and step, two 64 KiB stages, chunk swizzle (row>>1)&7 applied on the DMA source
// address).  What changes is the schedule inside the workgroup:
//
//   * a wave pulls ALL of its fragments of one K step into registers (16 A + 8 W
//     ds_read_b128 = 96 VGPRs) and then issues its 32 MFMAs from registers only;
//   * the 8 waves form two groups (wave>>2; waves w and w+4 share a SIMD) that run half a
//     K step apart: while one group multiplies (1024 matrix-pipe cycles per wave), the
//     other one reads its next fragments, so every SIMD always has one wave inside an
//     MFMA cluster — the matrix pipe never waits for LDS;
//   * because a stage is dead as soon as both groups have copied it to registers, the DMA
//     for K step t+1 is issued a full step before it is needed and drains with a plain
//     vmcnt(0) — exactly one DMA batch is in flight, always under MFMA work.
//
// One raw s_barrier per half step ("tick") keeps the two groups in phase.  Tick 2s issues
// the DMA of K step s+1; group 0 reads step t in tick 2t and multiplies in tick 2t+1,
// group 1 reads in tick 2t+1 and multiplies in tick 2t+2; every wave drains its DMA share
// at the end of the odd ticks.  The K reduction order is fixed (row results do not depend
// on M or on the position of the row).
// EXPERIMENT (not part of libzett_hip.so): built only by tools/gemm_bench.hip; measured
// results and why the product does not use it are in DESIGN.md §4.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm256.hip.h"

namespace zett {

// barrier that the compiler may not move memory operations across; LDS reads of this wave
// have returned (their stage may be overwritten by DMA after the barrier)
template <int ABL = 0>
__device__ __forceinline__ void tick_barrier() {
    if (ABL == 6) { __builtin_amdgcn_sched_barrier(0); return; }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int ABL = 0, int PF = 0>
__device__ __forceinline__ void tick_barrier_drain_dma() {
    if (PF > 0) { asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); return; }
    if (ABL == 6) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); return; }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// ABL: 0 = product kernel; 1 = ablation without DMA after the first two K steps; 2 = ablation without MFMAs
template <typename T, int ABL = 0, int PF = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm256p_tn_kernel(GemmArgs<T> g) {
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

    const unsigned char* a_src[4];
    const unsigned char* w_src[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = wave * 32 + j * 8 + (lane >> 3);
        const int ch = (lane & 7) ^ ((row >> 1) & 7);
        int ar = m0 + row; ar = ar < g.M ? ar : g.M - 1;
        int wr = n0 + row; wr = wr < g.N ? wr : g.N - 1;
        a_src[j] = (const unsigned char*)(g.A + (size_t)ar * g.lda) + ch * 16;
        w_src[j] = (const unsigned char*)(g.W + (size_t)wr * g.ldw) + ch * 16;
    }
    const int dma_base = wave * 32 * GEMM_ROW_BYTES;
    const int nk_total = g.K / BK;

    u32x4 wdummy[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wdummy[j] = u32x4{0, 0, 0, 0};
    auto issue_stage = [&](int kt) {
        if ((ABL == 1 || ABL == 5 || ABL == 6) && kt >= 2) return;
        if (ABL == 3) {   // ablation: twice the DMA traffic per step (second copy re-reads the previous K slice)
            unsigned char* sa2 = smem + (kt & 1) * G256_STAGE_BYTES + dma_base;
            unsigned char* sw2 = sa2 + G256_OPERAND_BYTES;
            const size_t koff2 = (size_t)(kt > 0 ? kt - 1 : 0) * GEMM_ROW_BYTES;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a_src[j] + koff2), (lds_ptr_t)(sa2 + j * 8 * GEMM_ROW_BYTES), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(w_src[j] + koff2), (lds_ptr_t)(sw2 + j * 8 * GEMM_ROW_BYTES), 16, 0, 0);
            }
        }
        unsigned char* sa = smem + (kt & 1) * G256_STAGE_BYTES + dma_base;
        unsigned char* sw = sa + G256_OPERAND_BYTES;
        const size_t koff = (size_t)kt * GEMM_ROW_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a_src[j] + koff), (lds_ptr_t)(sa + j * 8 * GEMM_ROW_BYTES), 16, 0, 0);
            if (ABL == 9) asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(wdummy[j]) : "v"(w_src[j] + koff));
            if (ABL != 4 && ABL != 7 && ABL != 9) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(w_src[j] + koff), (lds_ptr_t)(sw + j * 8 * GEMM_ROW_BYTES), 16, 0, 0);
        }
    };

    // L2 warm-up (PF > 0): the tiles of one XCD that share an A panel (same tm) or a W panel
    // (same tn) run side by side, so each of them pulls a distinct slice of the shared
    // panels into that XCD's L2 PF steps ahead: A rows [64*(tn&3), +64) and W rows
    // [32*(tm&7), +32), one dword per 128-byte line, 12 lanes per wave, as a 4-byte LDS-DMA
    // into a scratch area (no VGPR destination; it counts as one more VMEM op per step).
    const unsigned char* pf_src = nullptr;
    if (PF > 0) {
        const int idx = wave * 12 + lane;                    // 0..95 for lanes < 12
        int prow;
        const unsigned char* base;
        size_t ld;
        if (idx < 64) { prow = m0 + 64 * (tn & 3) + idx; prow = prow < g.M ? prow : g.M - 1; base = (const unsigned char*)g.A; ld = (size_t)g.lda * sizeof(T); }
        else { prow = n0 + 32 * (tm & 7) + (idx - 64); prow = prow < g.N ? prow : g.N - 1; base = (const unsigned char*)g.W; ld = (size_t)g.ldw * sizeof(T); }
        pf_src = base + (size_t)prow * ld;
    }
    auto prefetch_step = [&](int kt) {
        if (PF > 0 && lane < 12) {
            const int kc = kt < nk_total ? kt : nk_total - 1;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(pf_src + (size_t)kc * GEMM_ROW_BYTES),
                                             (lds_ptr_t)(smem + G256_LDS_BYTES + wave * 256), 4, 0, 0);
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addresses: A rows wm*128 + i*32 + l31, W rows wn*64 + j*32 + l31 of the stage image
    int a_off[4], w_off[2];
    int swz[4];   // swizzled chunk byte offsets for kk = 0..3 are (2*kk + hi) ^ s where s depends on the row only
#pragma unroll
    for (int i = 0; i < 4; ++i) a_off[i] = (wm * 128 + i * 32 + l31) * GEMM_ROW_BYTES;
#pragma unroll
    for (int j = 0; j < 2; ++j) w_off[j] = G256_OPERAND_BYTES + (wn * 64 + j * 32 + l31) * GEMM_ROW_BYTES;
    const int row_swz = (l31 >> 1) & 7;      // (row>>1)&7: the row bases are multiples of 32, so only l31 matters
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) swz[kk] = ((kk * 2 + hi) ^ row_swz) << 4;

    u32x4 fa[4][4], fw[4][2];     // [kk][i], [kk][j]

    auto read_frags = [&](int kt) {
        if ((ABL == 5 || ABL == 6) && kt > 0) return;
        const unsigned char* S = smem + (kt & 1) * G256_STAGE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int j = 0; j < 2; ++j) fw[kk][j] = *(const u32x4*)(S + w_off[j] + swz[kk]);
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[kk][i] = *(const u32x4*)(S + a_off[i] + swz[kk]);
        }
    };
    auto multiply = [&]() {
        if (ABL == 2 || ABL == 3 || ABL == 4) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(fa[kk][i]));
#pragma unroll
                for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(fw[kk][j]));
            }
            return;
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mfma_chunk<T>(fa[kk][i], fw[kk][j], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
    };

    const int nk = g.K / BK;
    issue_stage(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");                         // K step 0 is in LDS
    if (wm == 0) {
        for (int t = 0; t < nk; ++t) {
            if (t + 1 < nk) issue_stage(t + 1);       // tick 2t
            prefetch_step(t + 1 + PF);
            read_frags(t);
            tick_barrier<ABL>();
            multiply();                               // tick 2t+1
            tick_barrier_drain_dma<ABL, PF>();
        }
        tick_barrier<ABL>();                               // tick 2nk (group 1 multiplies)
    } else {
        if (1 < nk) issue_stage(1);                   // tick 0
        prefetch_step(1 + PF);
        tick_barrier<ABL>();
        for (int t = 0; t < nk; ++t) {
            read_frags(t);                            // tick 2t+1
            tick_barrier_drain_dma<ABL, PF>();
            if (t + 2 < nk) issue_stage(t + 2);       // tick 2t+2
            prefetch_step(t + 2 + PF);
            multiply();
            tick_barrier<ABL>();
        }
    }

    if (ABL == 9) {
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(wdummy[j]));
    }
    // ---- epilogue: identical to gemm256 (accumulators through a private 16 KiB LDS region per wave)
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

template <typename T, int ABL = 0, int PF = 0>
inline hipError_t launch_gemm256p(const GemmArgs<T>& g, hipStream_t stream) {
    const int tiles_m = (g.M + G256_BM - 1) / G256_BM;
    const int tiles_n = (g.N + G256_BN - 1) / G256_BN;
    if (tiles_m <= 0 || tiles_n <= 0) return hipSuccess;
    hipLaunchKernelGGL((gemm256p_tn_kernel<T, ABL, PF>), dim3(tiles_m * tiles_n), dim3(512), G256_LDS_BYTES + (PF > 0 ? 2048 : 0), stream, g);
    return hipGetLastError();
}

}  // namespace zett
