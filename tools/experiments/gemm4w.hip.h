// gemm4w.hip.h — experiment: 256x256 tile computed by FOUR waves (one per SIMD), each owning a
// 128x128 quadrant = 4x4 MFMA tiles of 32x32 (256 accumulator registers; the unified 512-entry
// register file of gfx950 holds them at one wave per SIMD).
//
// Why: per 16-byte K chunk a wave reads 4 A + 4 W fragments for 16 MFMAs (0.5 ds_read_b128 per
// MFMA) against 6 for 8 in the eight-wave kernel (0.75): a third less LDS read traffic competing
// with the LDS-DMA writes.  With a single wave per SIMD nothing hides a stall, so the K loop is
// software-pipelined by hand: the barrier of a K step sits between the third and the fourth
// MFMA group, the first fragments of the next step are read under the fourth group, and the
// DMA of the step after next is issued there as well.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "gemm256.hip.h"

namespace zett {

#ifndef G4W_SCHED
#define G4W_SCHED 1
#endif

__device__ unsigned long long g4w_trace[4096 * 4 * 8];    // [block][wave]{loop, vmcnt wait, barrier wait, steps} (G4W_TRACE builds)

template <typename T, int ACT = ACT_NONE, bool RES = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm4w_tn_kernel(GemmArgs<T> g) {
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
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // 0..3
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;

    // LDS-DMA plan: an operand image is 256 rows x 128 B = 32 wave instructions of 8 rows; wave w
    // issues instructions j = 0..7 for rows w*64 + j*8 + lane/8.
    const unsigned char* a_src[8];
    const unsigned char* w_src[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = wave * 64 + j * 8 + (lane >> 3);
        const int ch = (lane & 7) ^ ((row >> 1) & 7);
        int ar = m0 + row; ar = ar < g.M ? ar : g.M - 1;
        int wr = n0 + row; wr = wr < g.N ? wr : g.N - 1;
        a_src[j] = (const unsigned char*)(g.A + (size_t)ar * g.lda) + ch * 16;
        w_src[j] = (const unsigned char*)(g.W + (size_t)wr * g.ldw) + ch * 16;
    }
    const int dma_base = wave * 64 * GEMM_ROW_BYTES;
    auto dma_piece = [&](int kt, int stage, int j) {      // one A and one W instruction
        unsigned char* sa = smem + stage * G256_STAGE_BYTES + dma_base + j * 8 * GEMM_ROW_BYTES;
        const size_t koff = (size_t)kt * GEMM_ROW_BYTES;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a_src[j] + koff), (lds_ptr_t)sa, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(w_src[j] + koff), (lds_ptr_t)(sa + G256_OPERAND_BYTES), 16, 0, 0);
    };

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment offsets: row*128 + ((chunk ^ swz) << 4); 32-row steps leave swz unchanged
    const int swz = (l31 >> 1) & 7;
    int a_off[4], w_off[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int c = ((kk * 2 + hi) ^ swz) << 4;
        a_off[kk] = (wm * 128 + l31) * GEMM_ROW_BYTES + c;
        w_off[kk] = G256_OPERAND_BYTES + (wn * 128 + l31) * GEMM_ROW_BYTES + c;
    }
    u32x4 fa[2][4], fw[2][4];
    auto read_frags = [&](int stage, int kk, int set) {
        const unsigned char* S = smem + stage * G256_STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[set][i] = *(const u32x4*)(S + a_off[kk] + i * 32 * GEMM_ROW_BYTES);
#pragma unroll
        for (int j = 0; j < 4; ++j) fw[set][j] = *(const u32x4*)(S + w_off[kk] + j * 32 * GEMM_ROW_BYTES);
    };
    auto mfma_group = [&](int set) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) mfma_chunk<T>(fa[set][i], fw[set][j], acc[i][j]);
    };

    const int nk = g.K / BK;
    // ---- prologue: steps 0 and 1 in flight, step 0 landed, fragments of (0, kk=0) in set 0
#pragma unroll
    for (int j = 0; j < 8; ++j) dma_piece(0, 0, j);
    if (nk > 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) dma_piece(1, 1, j);
        __builtin_amdgcn_s_waitcnt(0x0F70 | 0x4000);     // vmcnt(16): the 16 requests of step 1 may still fly
    } else {
        __builtin_amdgcn_s_waitcnt(GEMM_WAIT_VMCNT0);
    }
    __builtin_amdgcn_s_barrier();
    read_frags(0, 0, 0);

    unsigned long long tr_v = 0, tr_b = 0;
    const unsigned long long tr_start = __builtin_readcyclecounter();
    const unsigned long long tr_rt0 = wall_clock64(); (void)tr_rt0;
    (void)tr_v; (void)tr_b; (void)tr_start;
    // one K step; MORE / MORE2 (compile-time, so that the step is one basic block the scheduler
    // can interleave) say whether steps kt+1 / kt+2 exist
    auto step = [&](int kt, auto more_c, auto more2_c) {
        constexpr bool more = decltype(more_c)::value, more2 = decltype(more2_c)::value;
        const int cur = kt & 1;
        // kk = 0..2: fragments of kk+1 are read under the MFMAs of kk
        read_frags(cur, 1, 1);
        mfma_group(0);
#if G4W_SCHED
#pragma unroll
        for (int q = 0; q < 8; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
#endif
        read_frags(cur, 2, 0);
        mfma_group(1);
#if G4W_SCHED
#pragma unroll
        for (int q = 0; q < 8; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
#endif
        read_frags(cur, 3, 1);
        mfma_group(0);
#if G4W_SCHED
#pragma unroll
        for (int q = 0; q < 8; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
#endif
        // every read of stage `cur` by this wave has completed (the fragments of kk = 3 are in
        // registers) and this wave's DMA requests for step kt+1 have landed
        __builtin_amdgcn_sched_barrier(0);
#ifdef G4W_TRACE
        __builtin_amdgcn_s_waitcnt(0xC07F);               // lgkmcnt(0) only
        const unsigned long long c0 = __builtin_readcyclecounter();
        __builtin_amdgcn_s_waitcnt(0x0070);
        const unsigned long long c1 = __builtin_readcyclecounter();
        __builtin_amdgcn_s_barrier();
        const unsigned long long c2 = __builtin_readcyclecounter();
        tr_v += c1 - c0; tr_b += c2 - c1;
#else
        __builtin_amdgcn_s_waitcnt(0x0070);               // vmcnt(0) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
#endif
        __builtin_amdgcn_sched_barrier(0);
        // kk = 3: first fragments of step kt+1 and the DMA of step kt+2 (into stage `cur`) under its MFMAs
        // (explicit micro-regions: 2 MFMAs, 1 fragment read, 1 A + 1 W DMA request each)
        {
            const unsigned char* S = smem + (cur ^ 1) * G256_STAGE_BYTES;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                mfma_chunk<T>(fa[1][(2 * q) >> 2], fw[1][(2 * q) & 3], acc[(2 * q) >> 2][(2 * q) & 3]);
                if (more) {
                    if (q < 4) fa[0][q] = *(const u32x4*)(S + a_off[0] + q * 32 * GEMM_ROW_BYTES);
                    else fw[0][q - 4] = *(const u32x4*)(S + w_off[0] + (q - 4) * 32 * GEMM_ROW_BYTES);
                }
                mfma_chunk<T>(fa[1][(2 * q + 1) >> 2], fw[1][(2 * q + 1) & 3], acc[(2 * q + 1) >> 2][(2 * q + 1) & 3]);
#if !defined(G4W_ABL) || G4W_ABL != 1
                if (more2) dma_piece(kt + 2, cur, q);
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    typedef std::integral_constant<bool, true> yes_t;
    typedef std::integral_constant<bool, false> no_t;
    int kt = 0;
    for (; kt + 2 < nk; ++kt) step(kt, yes_t{}, yes_t{});
    if (kt + 1 < nk) { step(kt, yes_t{}, no_t{}); ++kt; }
    step(kt, no_t{}, no_t{});
#ifdef G4W_TRACE
    if (lane == 0 && blockIdx.x < 4096) {
        unsigned long long* o = g4w_trace + (blockIdx.x * 4 + wave) * 8;
        o[0] = __builtin_readcyclecounter() - tr_start; o[1] = tr_v; o[2] = tr_b; o[3] = nk; o[4] = wall_clock64() - tr_rt0;
    }
#endif

    // ---- epilogue: each wave stages its 128x128 quadrant through a private 32 KiB LDS region
    // (64 rows x 128 fp32), two passes, drained as float4 per lane (two rows per instruction).
    __syncthreads();
    float* region = (float*)(smem + wave * 32768);
    const GemmEpilogue<T>& e = g.epi;
    const int c4 = l31 * 4;
    const int gcol = n0 + wn * 128 + c4;
    const bool col_ok = gcol < g.N;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f), sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = bias4;
    if (col_ok) {
        if (e.bias) bias4 = *(const float4*)(e.bias + gcol);
        if (e.scale) sc4 = *(const float4*)(e.scale + gcol);
        if (e.shift) sh4 = *(const float4*)(e.shift + gcol);
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        float4 o[32];
        if (RES) {
#pragma unroll
            for (int t = 0; t < 32; ++t) {
                const int grow = m0 + wm * 128 + p * 64 + t * 2 + hi;
                o[t] = (grow < g.M && col_ok) ? *(const float4*)(e.residual + (size_t)grow * e.ld_res + gcol) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    region[(i2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * 128 + j * 32 + l31] = acc[2 * p + i2][j][r];
        if (RES || p == 0) __builtin_amdgcn_s_waitcnt(GEMM_WAIT_VMCNT0);
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            const int lrow = t * 2 + hi;
            float4 v = *(const float4*)(region + lrow * 128 + c4);
            o[t] = epi_value4<ACT>(v, bias4, RES, RES ? o[t] : make_float4(0.f, 0.f, 0.f, 0.f), e.scale != nullptr, sc4, sh4);
        }
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            const int grow = m0 + wm * 128 + p * 64 + t * 2 + hi;
            if (grow >= g.M || !col_ok) continue;
            if (gcol < e.split_col) {
                if (e.out_f32) *(float4*)(e.out_f32 + (size_t)grow * e.ld_f32 + gcol) = o[t];
                if (e.out_lo) store_out4<T>(e.out_lo + (size_t)grow * e.ld_lo + gcol, o[t]);
            } else if (e.out_f32_b) {
                *(float4*)(e.out_f32_b + (size_t)grow * e.ld_f32 + (gcol - e.split_col)) = o[t];
            }
        }
    }
}

template <typename T, int ACT, bool RES>
inline hipError_t launch_gemm4w_inst(const GemmArgs<T>& g, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm4w_tn_kernel<T, ACT, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int tiles_m = (g.M + G256_BM - 1) / G256_BM;
    const int tiles_n = (g.N + G256_BN - 1) / G256_BN;
    if (tiles_m <= 0 || tiles_n <= 0) return hipSuccess;
    hipLaunchKernelGGL((gemm4w_tn_kernel<T, ACT, RES>), dim3(tiles_m * tiles_n), dim3(256), G256_LDS_BYTES, stream, g);
    return hipGetLastError();
}

template <typename T, int ACT>
inline hipError_t launch_gemm4w_act(const GemmArgs<T>& g, hipStream_t stream) {
    return g.epi.residual ? launch_gemm4w_inst<T, ACT, true>(g, stream) : launch_gemm4w_inst<T, ACT, false>(g, stream);
}

template <typename T>
inline hipError_t launch_gemm4w(const GemmArgs<T>& g, hipStream_t stream) {
    switch (g.epi.act) {
        case ACT_GELU_TANH: return launch_gemm4w_act<T, ACT_GELU_TANH>(g, stream);
        case ACT_GELU_ERF: return launch_gemm4w_act<T, ACT_GELU_ERF>(g, stream);
        default: return launch_gemm4w_act<T, ACT_NONE>(g, stream);
    }
}

}  // namespace zett
