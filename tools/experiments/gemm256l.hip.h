// gemm256l.hip.h — experiment: the 256x256 LDS-DMA kernel with DEDICATED LOADER WAVES.
//
// An LDS-DMA request (1 KiB per wave instruction) costs the issuing wave ~60-100 cycles
// (MI355X_MICROARCH.md, "LDS-DMA piece issue cost"; measured here: the four-wave experiment
// loses 830 cycles per K step to its 16 requests).  In gemm256.hip.h the eight compute waves
// issue the requests themselves, in lockstep, so the two waves of a SIMD tend to be stuck in
// a request at the same time.  Here the workgroup has 12 waves: waves 0-7 only read fragments
// and issue MFMAs (2(M) x 4(N), 128x64 each, as before), waves 8-11 (one per SIMD) only issue
// the LDS-DMA requests of the next K step, wait for them and join the per-step barrier.
// 3 waves per SIMD leave 168 VGPRs per wave.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm256.hip.h"

namespace zett {

template <typename T, int ACT = ACT_NONE, bool RES = false>
__global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(3, 3))) void gemm256l_tn_kernel(GemmArgs<T> g) {
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
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // 0..11
    const int nk = g.K / BK;

    if (wave >= 8) {
        // ---- loader wave lw: rows lw*64 + j*8 + lane/8 (j = 0..7) of both operand images
        const int lw = wave - 8;
        const unsigned char* a_src[8];
        const unsigned char* w_src[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = lw * 64 + j * 8 + (lane >> 3);
            const int ch = (lane & 7) ^ ((row >> 1) & 7);
            int ar = m0 + row; ar = ar < g.M ? ar : g.M - 1;
            int wr = n0 + row; wr = wr < g.N ? wr : g.N - 1;
            a_src[j] = (const unsigned char*)(g.A + (size_t)ar * g.lda) + ch * 16;
            w_src[j] = (const unsigned char*)(g.W + (size_t)wr * g.ldw) + ch * 16;
        }
        const int dma_base = lw * 64 * GEMM_ROW_BYTES;
        for (int kt = 0; kt <= nk; ++kt) {
            if (kt < nk) {
                unsigned char* sa = smem + (kt & 1) * G256_STAGE_BYTES + dma_base;
                const size_t koff = (size_t)kt * GEMM_ROW_BYTES;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a_src[j] + koff), (lds_ptr_t)(sa + j * 8 * GEMM_ROW_BYTES), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(w_src[j] + koff), (lds_ptr_t)(sa + G256_OPERAND_BYTES + j * 8 * GEMM_ROW_BYTES), 16, 0, 0);
                }
            }
            // barrier kt: step kt has landed (all loaders), the compute waves are done with step kt-1
            __builtin_amdgcn_s_waitcnt(GEMM_WAIT_VMCNT0);
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5;
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int a_row[4], w_row[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) a_row[i] = wm * 128 + i * 32 + l31;
#pragma unroll
    for (int j = 0; j < 2; ++j) w_row[j] = wn * 64 + j * 32 + l31;

    for (int kt = 0; kt < nk; ++kt) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                 // barrier kt
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* As = smem + (kt & 1) * G256_STAGE_BYTES;
        const unsigned char* Ws = As + G256_OPERAND_BYTES;
        u32x4 fa[2][4], fw[2][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[0][i] = *(const u32x4*)(As + lds_chunk_off(a_row[i], hi));
#pragma unroll
        for (int j = 0; j < 2; ++j) fw[0][j] = *(const u32x4*)(Ws + lds_chunk_off(w_row[j], hi));
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk < 3) {
                const int ch = (kk + 1) * 2 + hi;
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[nxt][i] = *(const u32x4*)(As + lds_chunk_off(a_row[i], ch));
#pragma unroll
                for (int j = 0; j < 2; ++j) fw[nxt][j] = *(const u32x4*)(Ws + lds_chunk_off(w_row[j], ch));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mfma_chunk<T>(fa[cur][i], fw[cur][j], acc[i][j]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                     // barrier nk: matches the loaders' last one
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue.  The accumulators go through LDS (free now) so that global traffic is
    // row-contiguous: each wave owns a private 16 KiB region = 64 rows x 64 fp32, filled from
    // the MFMA layout (col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) and drained
    // as float4 per lane, 16 lanes per 256-byte row segment.
    // (the loaders have exited; the eight compute waves passed barrier nk after their last reads)
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
        // The residual rows of this pass are requested first, all at once, so that their
        // latencies overlap each other and the LDS staging below.  The drain is split in two
        // straight-line phases — (A) LDS -> registers with bias/activation/residual/scale,
        // (B) nothing but stores — because loads and stores share the vmcnt counter: a wait for a
        // residual value placed between stores also waits for every earlier store to be
        // acknowledged, i.e. one full write latency per row group (measured: 11 us per tile).
        float4 o[16];
        if (RES) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int grow = m0 + wm * 128 + p * 64 + t * 4 + (lane >> 4);
                o[t] = (grow < g.M && col_ok) ? *(const float4*)(e.residual + (size_t)grow * e.ld_res + gcol) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    region[(i2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * 64 + j * 32 + l31] = acc[2 * p + i2][j][r];
        // every load of the epilogue (bias/scale/shift, this pass's residual rows) is complete
        // from here on: the compiler then needs no vmcnt wait inside the store sequence
        if (RES || p == 0) __builtin_amdgcn_s_waitcnt(GEMM_WAIT_VMCNT0);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int lrow = t * 4 + (lane >> 4);
            float4 v = *(const float4*)(region + lrow * 64 + c4);
            o[t] = epi_value4<ACT>(v, bias4, RES, RES ? o[t] : make_float4(0.f, 0.f, 0.f, 0.f), e.scale != nullptr, sc4, sh4);
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int grow = m0 + wm * 128 + p * 64 + t * 4 + (lane >> 4);
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
inline hipError_t launch_gemm256l_inst(const GemmArgs<T>& g, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm256l_tn_kernel<T, ACT, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int tiles_m = (g.M + G256_BM - 1) / G256_BM;
    const int tiles_n = (g.N + G256_BN - 1) / G256_BN;
    if (tiles_m <= 0 || tiles_n <= 0) return hipSuccess;
    hipLaunchKernelGGL((gemm256l_tn_kernel<T, ACT, RES>), dim3(tiles_m * tiles_n), dim3(768), G256_LDS_BYTES, stream, g);
    return hipGetLastError();
}

template <typename T, int ACT>
inline hipError_t launch_gemm256l_act(const GemmArgs<T>& g, hipStream_t stream) {
    return g.epi.residual ? launch_gemm256l_inst<T, ACT, true>(g, stream) : launch_gemm256l_inst<T, ACT, false>(g, stream);
}

template <typename T>
inline hipError_t launch_gemm256l(const GemmArgs<T>& g, hipStream_t stream) {
    switch (g.epi.act) {
        case ACT_GELU_TANH: return launch_gemm256l_act<T, ACT_GELU_TANH>(g, stream);
        case ACT_GELU_ERF: return launch_gemm256l_act<T, ACT_GELU_ERF>(g, stream);
        default: return launch_gemm256l_act<T, ACT_NONE>(g, stream);
    }
}

}  // namespace zett
