// gemm256.hip.h — the LDS-DMA 256x256 MFMA GEMM for gfx950 (tile variant 5: the round's first large-tile
// kernel, since superseded as the default by the register-staged gemm8r.hip.h / gemm8x.hip.h, which are
// 7-11 % faster) (the epilogue drain shared by every large tile, EpiDrain, lives in zett_amd/csrc/gemm_tile.hip.h).  256x256 output tile, 8 waves,
// K staged 128 bytes per row and step straight from HBM/L2 into LDS by the LDS-DMA path
// (global_load_lds_dwordx4: no VGPR round trip, no ds_write pass), two LDS stages of
// 64 KiB so the DMA of K-step t+1 runs under the MFMAs of K-step t, one barrier per step.
//
//   C[M,N] = epilogue( A[M,K] · W[N,K]ᵀ )      (same contract and epilogue as gemm.hip.h)
//
// LDS image per operand and stage: 256 rows x 128 B, 16-byte chunks XOR-swizzled by
// (row>>1)&7.  LDS-DMA writes are lane-linear (wave-uniform base + lane*16), so the swizzle
// is applied to each lane's SOURCE address (the 8 lanes of a row permute the 8 chunks of
// one 128-byte line: still one coalesced line per row) and undone by the same XOR on the
// ds_read_b128 side.
//
// Waves are laid out 2 (M) x 4 (N); a wave owns 128x64 of the tile = 4x2 MFMA tiles of
// 32x32 (128 accumulator registers).  Per 16-byte K chunk a wave issues 6 ds_read_b128
// (4 A, 2 W) for 8 v_mfma_f32_32x32x16_bf16 (or 32 v_mfma_f32_32x32x2_f32 in fp32 mode).
// The reduction order over K is fixed, so a row's result does not depend on M or on
// where the row sits in the launch.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_tile.hip.h"

namespace zett {

// VAR (experiments, tools/gemm_bench): bit 0 = spread the DMA issue over the 4 K chunks of a step,
// bit 1 = s_setprio(1) around the MFMA groups.  The library instantiates VAR = 1.
template <typename T, int VAR = 0, int ACT = ACT_NONE, bool RES = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm256_tn_kernel(GemmArgs<T> g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BK = GEMM_ROW_BYTES / (int)sizeof(T);

    const int tiles_m = (g.M + G256_BM - 1) / G256_BM;
    const int tiles_n = (g.N + G256_BN - 1) / G256_BN;
    const int nwg = tiles_m * tiles_n;
    int wg = blockIdx.x;
    {   // XCD-aware order (block b runs on XCD b % 8): give each XCD a contiguous tile range
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

    // LDS-DMA plan: a wave instruction moves 64 x 16 B = 8 rows of 128 B.  The 8 waves x 4
    // instructions cover the 256 rows of one operand; lane l of instruction j of wave w
    // fills physical slot (row = w*32 + j*8 + l/8, chunk = l%8) from logical chunk
    // (l%8) ^ swz(row).
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
    const int dma_base = wave * 32 * GEMM_ROW_BYTES;    // byte offset of this wave's 32 rows in an operand image

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

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read offsets (bytes) inside an operand image, without the chunk term
    int a_row[4], w_row[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) a_row[i] = wm * 128 + i * 32 + l31;
#pragma unroll
    for (int j = 0; j < 2; ++j) w_row[j] = wn * 64 + j * 32 + l31;

    const int nk = g.K / BK;
    if ((VAR & 16) && blockIdx.x < 256) {
        // experiment: skew the first round of workgroups by eighths of a tile time so that the
        // store bursts of the epilogues of different CUs do not coincide
        const int delay = ((blockIdx.x >> 3) & 7) * nk * 5;      // in units of 64 cycles: a tile is ~nk*2600 cycles
        for (int i = 0; i < delay; ++i) __builtin_amdgcn_s_sleep(1);
    }
    issue_stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        // every wave's DMA for step kt has landed (the barrier carries vmcnt(0)) and every
        // wave is done reading the other stage
        __syncthreads();
        const bool more = kt + 1 < nk;
        if (more && !(VAR & 1)) issue_stage(kt + 1, (kt + 1) & 1);
        const unsigned char* As = smem + (kt & 1) * G256_STAGE_BYTES;
        const unsigned char* Ws = As + G256_OPERAND_BYTES;
        // register double-buffered fragments: the ds_reads of chunk kk+1 are in flight under
        // the 8 MFMAs of chunk kk
        u32x4 fa[2][4], fw[2][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[0][i] = *(const u32x4*)(As + lds_chunk_off(a_row[i], hi));
#pragma unroll
        for (int j = 0; j < 2; ++j) fw[0][j] = *(const u32x4*)(Ws + lds_chunk_off(w_row[j], hi));
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if ((VAR & 1) && more) {   // one quarter of the next stage per chunk: A piece kk and W piece kk
                unsigned char* sa = smem + ((kt + 1) & 1) * G256_STAGE_BYTES + dma_base;
                const size_t koff = (size_t)(kt + 1) * GEMM_ROW_BYTES;
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a_src[kk] + koff), (lds_ptr_t)(sa + kk * 8 * GEMM_ROW_BYTES), 16, 0, (VAR & 4) ? 2 : 0);
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(w_src[kk] + koff), (lds_ptr_t)(sa + G256_OPERAND_BYTES + kk * 8 * GEMM_ROW_BYTES), 16, 0, (VAR & 8) ? 2 : 0);
            }
            if (kk < 3) {
                const int ch = (kk + 1) * 2 + hi;
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[nxt][i] = *(const u32x4*)(As + lds_chunk_off(a_row[i], ch));
#pragma unroll
                for (int j = 0; j < 2; ++j) fw[nxt][j] = *(const u32x4*)(Ws + lds_chunk_off(w_row[j], ch));
            }
            if (VAR & 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mfma_chunk<T>(fa[cur][i], fw[cur][j], acc[i][j]);
            if (VAR & 2) __builtin_amdgcn_s_setprio(0);
        }
    }

    // ---- epilogue.  The accumulators go through LDS (free now) so that global traffic is
    // row-contiguous: each wave owns a private 16 KiB region = 64 rows x 64 fp32, filled from
    // the MFMA layout (col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) and drained by
    // EpiDrain, eight columns per lane.
    __syncthreads();                                  // all waves are done with the K stages
    float* region = (float*)(smem + wave * 16384);
    typedef EpiDrain<T, ACT, RES, 64, 64> Drain;
    const int gcol = n0 + wn * 64 + (lane % Drain::LPR) * 8;
    const bool col_ok = gcol < g.N;
    float4 bias8[2], sc8[2], sh8[2];
    Drain::load_cols(g.epi, gcol, col_ok, bias8, sc8, sh8);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        float4 oa[Drain::NIT], ob[Drain::NIT];
        const int row0 = m0 + wm * 128 + p * 64;
        Drain::load_res(g, row0, gcol, col_ok, lane, oa, ob);
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    region[(i2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * 64 + j * 32 + l31] = acc[2 * p + i2][j][r];
        if (RES || p == 0) __builtin_amdgcn_s_waitcnt(GEMM_WAIT_VMCNT0);
        Drain::drain(g, region, row0, gcol, col_ok, lane, bias8, sc8, sh8, oa, ob);
    }
}

template <typename T, int VAR, int ACT, bool RES>
inline hipError_t launch_gemm256_inst(const GemmArgs<T>& g, hipStream_t stream) {
    static bool attr_set = false;     // > 64 KiB of dynamic LDS needs the attribute once per instantiation
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm256_tn_kernel<T, VAR, ACT, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int tiles_m = (g.M + G256_BM - 1) / G256_BM;
    const int tiles_n = (g.N + G256_BN - 1) / G256_BN;
    if (tiles_m <= 0 || tiles_n <= 0) return hipSuccess;
    hipLaunchKernelGGL((gemm256_tn_kernel<T, VAR, ACT, RES>), dim3(tiles_m * tiles_n), dim3(512), G256_LDS_BYTES, stream, g);
    return hipGetLastError();
}

template <typename T, int VAR, int ACT>
inline hipError_t launch_gemm256_act(const GemmArgs<T>& g, hipStream_t stream) {
    return g.epi.residual ? launch_gemm256_inst<T, VAR, ACT, true>(g, stream) : launch_gemm256_inst<T, VAR, ACT, false>(g, stream);
}

template <typename T, int VAR = 0>
inline hipError_t launch_gemm256(const GemmArgs<T>& g, hipStream_t stream) {
    switch (g.epi.act) {
        case ACT_GELU_TANH: return launch_gemm256_act<T, VAR, ACT_GELU_TANH>(g, stream);
        case ACT_GELU_ERF: return launch_gemm256_act<T, VAR, ACT_GELU_ERF>(g, stream);
        default: return launch_gemm256_act<T, VAR, ACT_NONE>(g, stream);
    }
}

}  // namespace zett
