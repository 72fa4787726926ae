// gemm384.hip.h — 384x256 MFMA GEMM tile, 12 waves (3 per SIMD), for gfx950.
//
//   C[M,N] = epilogue( A[M,K] · W[N,K]ᵀ )      (contract and epilogue of gemm.hip.h)
//
// Why: the 256x256 kernel is bound by the L2->LDS DMA rate per CU (halving
// the staged bytes lifts it from ~1.13 to ~1.44 PFLOP/s, tools/gemm_bench ABL=1).  A 384x256
// tile stages (384+256)*128 B = 80 KiB per K step for 1.5x the FLOPs of the 256x256 tile:
// 17 % fewer staged bytes per FLOP.  The accumulators of the tile fill 3/4 of a SIMD's
// register file (3 waves x 128 VGPRs), which is as far as the tile can grow.
//
// Same structure as gemm256.hip.h: LDS-DMA (global_load_lds_dwordx4) into two stages of
// 80 KiB (all 160 KiB of LDS), 128-byte rows with the source-side XOR swizzle (row>>1)&7,
// one barrier per K step, DMA requests of the next step spread over the four 16-byte K
// chunks, waves 3(M) x 4(N) each owning 128x64 = 4x2 MFMA tiles of 32x32.  The fixed K
// reduction order makes results bit-identical to the other GEMM kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_tile.hip.h"

namespace zett {

constexpr int G384_BM = 384;
constexpr int G384_BN = 256;
constexpr int G384_A_BYTES = G384_BM * GEMM_ROW_BYTES;            // 48 KiB
constexpr int G384_W_BYTES = G384_BN * GEMM_ROW_BYTES;            // 32 KiB
constexpr int G384_STAGE_BYTES = G384_A_BYTES + G384_W_BYTES;     // 80 KiB
constexpr int G384_LDS_BYTES = 2 * G384_STAGE_BYTES;              // 160 KiB

template <typename T, int ACT = ACT_NONE, bool RES = false>
__global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(3, 3))) void gemm384_tn_kernel(GemmArgs<T> g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BK = GEMM_ROW_BYTES / (int)sizeof(T);

    const int tiles_m = (g.M + G384_BM - 1) / G384_BM;
    const int tiles_n = (g.N + G384_BN - 1) / G384_BN;
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
    const int m0 = tm * G384_BM, n0 = tn * G384_BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // 0..11
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5;

    // LDS-DMA plan: a wave instruction moves 8 rows x 128 B.  A: 48 instructions = 4 per wave
    // (rows wave*32 + j*8 + lane/8).  W: 32 instructions = 4 per wave for waves 0..7.
    // Register diet (3 waves per SIMD leave 168 VGPRs): no per-row clamping — the caller
    // guarantees that A has at least tiles_m*384 readable rows and N is a multiple of 256 — so
    // the four source rows of a wave differ by constant strides; the source-side swizzle
    // (lane&7) ^ ((row>>1)&7) only depends on the parity of j, hence two base pointers per operand.
    const bool has_w = wave < 8;
    const unsigned char* a_base[2];
    const unsigned char* w_base[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int row = wave * 32 + par * 8 + (lane >> 3);
        const int ch = (lane & 7) ^ ((row >> 1) & 7);
        a_base[par] = (const unsigned char*)(g.A + (size_t)(m0 + row) * g.lda) + ch * 16;
        w_base[par] = (const unsigned char*)(g.W + (size_t)(n0 + (has_w ? row : 0)) * g.ldw) + ch * 16;
    }
    const size_t a_stride16 = (size_t)16 * g.lda * sizeof(T), w_stride16 = (size_t)16 * g.ldw * sizeof(T);
    const int dma_base = wave * 32 * GEMM_ROW_BYTES;

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

    auto issue_piece = [&](int kt, int piece) {
        unsigned char* sa = smem + (kt & 1) * G384_STAGE_BYTES + dma_base;
        const size_t koff = (size_t)kt * GEMM_ROW_BYTES;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a_base[piece & 1] + (piece >> 1) * a_stride16 + koff),
                                         (lds_ptr_t)(sa + piece * 8 * GEMM_ROW_BYTES), 16, 0, 0);
        if (has_w)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(w_base[piece & 1] + (piece >> 1) * w_stride16 + koff),
                                             (lds_ptr_t)(sa + G384_A_BYTES + piece * 8 * GEMM_ROW_BYTES), 16, 0, 0);
    };

    const int nk = g.K / BK;
#pragma unroll
    for (int piece = 0; piece < 4; ++piece) issue_piece(0, piece);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();               // carries vmcnt(0): step kt has landed; the other stage is dead
        const bool more = kt + 1 < nk;
        const unsigned char* As = smem + (kt & 1) * G384_STAGE_BYTES;
        const unsigned char* Ws = As + G384_A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int ch = kk * 2 + hi;
            if (more) issue_piece(kt + 1, kk);
            u32x4 fa[4], fw[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = *(const u32x4*)(As + lds_chunk_off(a_row[i], ch));
#pragma unroll
            for (int j = 0; j < 2; ++j) fw[j] = *(const u32x4*)(Ws + lds_chunk_off(w_row[j], ch));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mfma_chunk<T>(fa[i], fw[j], acc[i][j]);
        }
    }

    // ---- epilogue: accumulators through a private 8 KiB LDS region per wave (32 rows x 64 fp32),
    // four passes (one per 32-row MFMA tile row), drained by EpiDrain (gemm_tile.hip.h).
    __syncthreads();
    float* region = (float*)(smem + wave * 8192);
    typedef EpiDrain<T, ACT, RES, 32, 64, false, true, false> Drain;      // no scale/shift: launch_gemm384 refuses such epilogues
    const int gcol = n0 + wn * 64 + (lane % Drain::LPR) * 8;
    const bool col_ok = gcol < g.N;
    float4 bias8[2], sc8[2], sh8[2];
    Drain::load_cols(g.epi, gcol, col_ok, bias8, sc8, sh8);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float4 oa[Drain::NIT], ob[Drain::NIT];
        const int row0 = m0 + wm * 128 + i * 32;
        Drain::load_res(g, row0, gcol, col_ok, lane, oa, ob);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                region[((r & 3) + 8 * (r >> 2) + 4 * hi) * 64 + j * 32 + l31] = acc[i][j][r];
        if (RES || i == 0) __builtin_amdgcn_s_waitcnt(GEMM_WAIT_VMCNT0);
        Drain::drain(g, region, row0, gcol, col_ok, lane, bias8, sc8, sh8, oa, ob);
    }
}

template <typename T, int ACT, bool RES>
inline hipError_t launch_gemm384_inst(const GemmArgs<T>& g, hipStream_t stream) {
    static DeviceFlags attr;      // per device: one process may hold a handle per GPU
    bool* done = attr.current();
    if (!done || !*done) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm384_tn_kernel<T, ACT, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, G384_LDS_BYTES);
        if (e != hipSuccess) return e;
        if (done) *done = true;
    }
    if (g.epi.scale || g.epi.shift) return hipErrorInvalidValue;      // register budget: EpiDrain<..., SCALE = false>
    const int tiles_m = (g.M + G384_BM - 1) / G384_BM;
    const int tiles_n = (g.N + G384_BN - 1) / G384_BN;
    if (tiles_m <= 0 || tiles_n <= 0) return hipSuccess;
    hipLaunchKernelGGL((gemm384_tn_kernel<T, ACT, RES>), dim3(tiles_m * tiles_n), dim3(768), G384_LDS_BYTES, stream, g);
    return hipGetLastError();
}

// Residual epilogues are refused: at 168 VGPRs per wave the residual drain spills to scratch, and a
// spilled epilogue of this kernel returned wrong tiles on the first launches of a process (f16,
// N = 768; tools/gemm_bench STRESS runs) — no instantiation with scratch is ever launched.
template <typename T, int ACT>
inline hipError_t launch_gemm384_act(const GemmArgs<T>& g, hipStream_t stream) {
    if (g.epi.residual) return hipErrorInvalidValue;
    return launch_gemm384_inst<T, ACT, false>(g, stream);
}

template <typename T>
inline hipError_t launch_gemm384(const GemmArgs<T>& g, hipStream_t stream) {
    switch (g.epi.act) {
        case ACT_GELU_TANH: return launch_gemm384_act<T, ACT_GELU_TANH>(g, stream);
        case ACT_GELU_ERF: return launch_gemm384_act<T, ACT_GELU_ERF>(g, stream);
        default: return launch_gemm384_act<T, ACT_NONE>(g, stream);
    }
}

}  // namespace zett
