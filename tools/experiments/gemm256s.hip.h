// gemm256s.hip.h — experiment: PERSISTENT form of the 256x256 LDS-DMA kernel (gemm256.hip.h).
//
// One workgroup per CU walks a list of output tiles.  The K-step stream is continuous across
// tile boundaries: the LDS-DMA of the next tile's first K step is issued before the epilogue of
// the current tile (into the stage the last step did not use), and the epilogue stages the
// accumulators through the stage buffer of the last step (8 KiB per wave, 4 passes).  This hides
// the per-tile prologue (one exposed DMA latency) and the workgroup launch gap, which matter
// when K is short (16 K-steps per tile for K = 1024).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm256.hip.h"

namespace zett {

__device__ unsigned long long g256s_trace[256 * 64 * 4];   // [block][tile][event] wall_clock64 stamps (TRACE builds)
#ifdef G256S_TRACE
#define G256S_STAMP(ev) do { if (tid == 0 && tix < 64) g256s_trace[(blockIdx.x * 64 + tix) * 4 + (ev)] = wall_clock64(); } while (0)
#else
#define G256S_STAMP(ev) do { } while (0)
#endif

__device__ int g256s_skew_ticks = 0;      // start delay per XCD index in 10 ns ticks (experiment)
__device__ int g256s_skew_mode = 0;       // 0: by XCD, 1: by CU slot inside the XCD

template <typename T, int ACT = ACT_NONE, bool RES = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm256s_tn_kernel(GemmArgs<T> g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BK = GEMM_ROW_BYTES / (int)sizeof(T);

    const int tiles_m = (g.M + G256_BM - 1) / G256_BM;
    const int tiles_n = (g.N + G256_BN - 1) / G256_BN;
    const int nwg = tiles_m * tiles_n;
    // XCD x (= blockIdx % 8) owns the contiguous tile range [x_first, x_first + x_count)
    const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3, li = blockIdx.x >> 3;
    const int q = nwg >> 3, r8 = nwg & 7;
    const int x_first = xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q;
    const int x_count = q + (xcd < r8 ? 1 : 0);
    constexpr int GROUP_M = 8;
    const int group_size = GROUP_M * tiles_n;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5;
    const int dma_base = wave * 32 * GEMM_ROW_BYTES;

    const unsigned char* a_src[4];
    const unsigned char* w_src[4];
    auto tile_origin = [&](int t, int& m0, int& n0) {
        const int wg = x_first + t;
        const int first_m = (wg / group_size) * GROUP_M;
        const int gm = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
        m0 = (first_m + (wg % group_size) % gm) * G256_BM;
        n0 = ((wg % group_size) / gm) * G256_BN;
    };
    auto set_sources = [&](int m0, int n0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = wave * 32 + j * 8 + (lane >> 3);
            const int ch = (lane & 7) ^ ((row >> 1) & 7);
            int ar = m0 + row; ar = ar < g.M ? ar : g.M - 1;
            int wr = n0 + row; wr = wr < g.N ? wr : g.N - 1;
            a_src[j] = (const unsigned char*)(g.A + (size_t)ar * g.lda) + ch * 16;
            w_src[j] = (const unsigned char*)(g.W + (size_t)wr * g.ldw) + ch * 16;
        }
    };
    auto issue_stage = [&](int kt, int stage) {
        unsigned char* sa = smem + stage * G256_STAGE_BYTES + dma_base;
        unsigned char* sw = sa + G256_OPERAND_BYTES;
        const size_t koff = (size_t)kt * GEMM_ROW_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a_src[j] + koff), (lds_ptr_t)(sa + j * 8 * GEMM_ROW_BYTES), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(w_src[j] + koff), (lds_ptr_t)(sw + j * 8 * GEMM_ROW_BYTES), 16, 0, 0);
        }
    };

    int a_row[4], w_row[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) a_row[i] = wm * 128 + i * 32 + l31;
#pragma unroll
    for (int j = 0; j < 2; ++j) w_row[j] = wn * 64 + j * 32 + l31;

    const int nk = g.K / BK;
    const GemmEpilogue<T>& e = g.epi;
    const int c4 = (lane & 15) * 4;

    int t = li;
    if (t >= x_count) return;
    int m0, n0;
    tile_origin(t, m0, n0);
    set_sources(m0, n0);
    int s = 0;                       // running K-step counter: stage of step (s) is s & 1
    if (g256s_skew_ticks > 0) {
        const unsigned long long until = wall_clock64() + (unsigned long long)((g256s_skew_mode ? (li & 7) : xcd) * g256s_skew_ticks);
        while (wall_clock64() < until) __builtin_amdgcn_s_sleep(8);
    }
    issue_stage(0, 0);
    int tix = 0;
    for (; t < x_count; t += per_xcd, ++tix) {
        G256S_STAMP(0);
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        for (int kt = 0; kt < nk; ++kt) {
            __syncthreads();
            const bool more = kt + 1 < nk;
            const unsigned char* As = smem + ((s + kt) & 1) * G256_STAGE_BYTES;
            const unsigned char* Ws = As + G256_OPERAND_BYTES;
            u32x4 fa[2][4], fw[2][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[0][i] = *(const u32x4*)(As + lds_chunk_off(a_row[i], hi));
#pragma unroll
            for (int j = 0; j < 2; ++j) fw[0][j] = *(const u32x4*)(Ws + lds_chunk_off(w_row[j], hi));
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int cur = kk & 1, nxt = cur ^ 1;
                if (more) {
                    unsigned char* sa = smem + ((s + kt + 1) & 1) * G256_STAGE_BYTES + dma_base;
                    const size_t koff = (size_t)(kt + 1) * GEMM_ROW_BYTES;
                    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a_src[kk] + koff), (lds_ptr_t)(sa + kk * 8 * GEMM_ROW_BYTES), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(w_src[kk] + koff), (lds_ptr_t)(sa + G256_OPERAND_BYTES + kk * 8 * GEMM_ROW_BYTES), 16, 0, 0);
                }
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
        s += nk;
        G256S_STAMP(1);

        // ---- tile boundary: every wave is done with both stages.  Start the next tile's first
        // K step into the stage the last step did not use, then drain the accumulators through
        // the last step's stage buffer.
        __syncthreads();
        G256S_STAMP(2);
        const int em0 = m0, en0 = n0;
        if (t + per_xcd < x_count) {
            tile_origin(t + per_xcd, m0, n0);
            set_sources(m0, n0);
            issue_stage(0, s & 1);
        }
        float* region = (float*)(smem + ((s - 1) & 1) * G256_STAGE_BYTES + wave * 8192);
        const int gcol = en0 + wn * 64 + c4;
        const bool col_ok = gcol < g.N;
        float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f), sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = bias4;
        if (col_ok) {
            if (e.bias) bias4 = *(const float4*)(e.bias + gcol);
            if (e.scale) sc4 = *(const float4*)(e.scale + gcol);
            if (e.shift) sh4 = *(const float4*)(e.shift + gcol);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float4 o[8];
            if (RES) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int grow = em0 + wm * 128 + p * 32 + u * 4 + (lane >> 4);
                    o[u] = (grow < g.M && col_ok) ? *(const float4*)(e.residual + (size_t)grow * e.ld_res + gcol) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    region[((r & 3) + 8 * (r >> 2) + 4 * hi) * 64 + j * 32 + l31] = acc[p][j][r];
            if (RES || p == 0) __builtin_amdgcn_s_waitcnt(GEMM_WAIT_VMCNT0);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int lrow = u * 4 + (lane >> 4);
                float4 v = *(const float4*)(region + lrow * 64 + c4);
                o[u] = epi_value4<ACT>(v, bias4, RES, RES ? o[u] : make_float4(0.f, 0.f, 0.f, 0.f), e.scale != nullptr, sc4, sh4);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int grow = em0 + wm * 128 + p * 32 + u * 4 + (lane >> 4);
                if (grow >= g.M || !col_ok) continue;
                if (gcol < e.split_col) {
                    if (e.out_f32) *(float4*)(e.out_f32 + (size_t)grow * e.ld_f32 + gcol) = o[u];
                    if (e.out_lo) store_out4<T>(e.out_lo + (size_t)grow * e.ld_lo + gcol, o[u]);
                } else if (e.out_f32_b) {
                    *(float4*)(e.out_f32_b + (size_t)grow * e.ld_f32 + (gcol - e.split_col)) = o[u];
                }
            }
        }
        G256S_STAMP(3);
    }
}

template <typename T, int ACT, bool RES>
inline hipError_t launch_gemm256s_inst(const GemmArgs<T>& g, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm256s_tn_kernel<T, ACT, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int tiles_m = (g.M + G256_BM - 1) / G256_BM;
    const int tiles_n = (g.N + G256_BN - 1) / G256_BN;
    const int nwg = tiles_m * tiles_n;
    if (nwg <= 0) return hipSuccess;
    const int grid = nwg >= 256 ? 256 : ((nwg + 7) / 8) * 8;
    hipLaunchKernelGGL((gemm256s_tn_kernel<T, ACT, RES>), dim3(grid), dim3(512), G256_LDS_BYTES, stream, g);
    return hipGetLastError();
}

template <typename T, int ACT>
inline hipError_t launch_gemm256s_act(const GemmArgs<T>& g, hipStream_t stream) {
    return g.epi.residual ? launch_gemm256s_inst<T, ACT, true>(g, stream) : launch_gemm256s_inst<T, ACT, false>(g, stream);
}

template <typename T>
inline hipError_t launch_gemm256s(const GemmArgs<T>& g, hipStream_t stream) {
    switch (g.epi.act) {
        case ACT_GELU_TANH: return launch_gemm256s_act<T, ACT_GELU_TANH>(g, stream);
        case ACT_GELU_ERF: return launch_gemm256s_act<T, ACT_GELU_ERF>(g, stream);
        default: return launch_gemm256s_act<T, ACT_NONE>(g, stream);
    }
}

}  // namespace zett
