// gemm4d.hip.h — tile variant 7: 256x256 tile, FOUR waves (one per SIMD, 128x128 of the tile each), both operands
// streamed HBM/L2 -> LDS by buffer_load_dwordx4 ... lds (no VGPR round trip, no ds_write pass), on
// v_mfma_f32_16x16x32_{bf16,f16} with the 256 accumulator registers pinned to AGPRs.  The tile of every 16-bit
// launch with K >= 2048: with the 4 x 8 tile order (ZETT_GROUP_M = 4) 4-8 % faster than gemm8x on the shapes of
// the benchmark step, residual or not (tools/gemm_bench), 5 % slower at K = 1024.  Same geometry as the hipBLASLt kernel the yardstick runs; the experiments that led
// here (ablations, K-start staggering, schedules with 3-4 barriers) are in tools/experiments/gemm4dx.hip.h.
//
//   C[M,N] = epilogue( A[M,K] · W[N,K]ᵀ )      (contract and epilogue of gemm.hip.h; bit-identical results)
//
// K step t of a wave = 128 MFMAs, K block 0 (64) then K block 1 (64); LDS stage t&1 holds step t (64 KiB).
//   p =   0..30   read the block-1 fragments of step t (8 W, 8 A)
//   p =  36       wait for them + barrier: both images of stage t&1 are free
//   p =  38..101  the 16 LDS-DMA requests of step t+2 into that stage, one per MFMA for the whole CU: wave w
//                 issues its request r under MFMA 38 + 4r + w (each wave runs its own copy of the loop), so the
//                 four waves never queue at the texture path at once — a blocked request blocks the MFMAs
//                 behind it, and with one wave per SIMD nothing else can fill the pipe
//   p = 102       vmcnt: everything requested during step t-1 (= step t+1) has landed + barrier
//   p = 103..125  read the block-0 fragments of step t+1
// Fragments are double-buffered in 128 VGPRs.  A wave reads 32 KiB of fragments per step for 128 MFMAs: a third
// less LDS traffic per FLOP than the eight-wave kernels, and no LDS write traffic from the waves.
//
// LDS image and swizzle as in gemm256.hip.h: per stage and operand 256 rows x 128 B, 16-byte chunks
// XOR-swizzled by (row>>1)&7, applied to each lane's SOURCE address and undone on the ds_read_b128 side.
// The K reduction order per accumulator is that of every other tile variant.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "gemm8x.hip.h"

namespace zett {

// s_waitcnt vmcnt(n) only (gfx9 encoding: vmcnt = simm16[15:14]:[3:0], expcnt [6:4], lgkmcnt [11:8])
constexpr int g4d_wait_vm(int n) { return ((n >> 4) << 14) | 0x0F70 | (n & 15); }

// The accumulators are pinned to the AGPR half of the register file ("+a") and the fragments to the VGPR
// half: with 256 + 128 live registers the allocator otherwise spreads the accumulators over both halves and
// shuttles them through v_accvgpr_read/write around every MFMA.  Nothing reads an accumulator between the
// MFMAs of the K loop (64 MFMAs lie between two uses of the same one); the caller covers the MFMA -> VALU
// read latency after the loop, which the hazard recogniser cannot see through inline assembly.
template <typename T> __device__ __forceinline__ void mfma16_agpr(f32x4& c, const u32x4& a, const u32x4& b);
template <> __device__ __forceinline__ void mfma16_agpr<bf16_t>(f32x4& c, const u32x4& a, const u32x4& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
template <> __device__ __forceinline__ void mfma16_agpr<f16_t>(f32x4& c, const u32x4& a, const u32x4& b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

template <typename T, int ACT = ACT_NONE, bool RES = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm4d_tn_kernel(GemmArgs<T> g) {
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
    // Tile order: each XCD owns a contiguous range of the order (remap above); the order walks the GROUP_N column tiles
    // of a group first, then steps one row tile down, so the 32 tiles an XCD runs at a time are 8 row tiles x 4 column
    // tiles and its next 32 are the next 8 row tiles of the SAME 4 column tiles: the W panels (the small operand, MALL-
    // resident) stay, the A panels stream.  Against row-major groups of 4 (what gemm8r/gemm8x use): +5 % on the
    // K = 8192 launches, equal on the others (tools/gemm_bench g4dv, G4DX_MAP / G4DX_GROUP_M sweep).
    constexpr int GROUP_N = 4;
    int tm, tn;
    if (g.tile_order == 0) {
        const int group_size = GROUP_N * tiles_m;
        const int first_n = (wg / group_size) * GROUP_N;
        const int gn = (tiles_n - first_n) < GROUP_N ? (tiles_n - first_n) : GROUP_N;
        tn = first_n + (wg % group_size) % gn;
        tm = (wg % group_size) / gn;
    } else {                      // zett_set_option("gemm_tile_order", 1): ZETT_GROUP_M row tiles first, as gemm8r / gemm8x
        constexpr int GROUP_M = ZETT_GROUP_M;
        const int group_size = GROUP_M * tiles_n;
        const int first_m = (wg / group_size) * GROUP_M;
        const int gm = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
        tm = first_m + (wg % group_size) % gm;
        tn = (wg % group_size) / gm;
    }
    const int m0 = tm * G256_BM, n0 = tn * G256_BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // 0..3
    const int wm = wave >> 1, wn = wave & 1;

    // request r (0..7) of a wave moves rows wave*64 + r*8 + lane/8 of an operand; the lane's LDS slot is
    // chunk lane%8 of its row, which holds source chunk (lane%8) ^ swz(row).  Rows past the edge are clamped.
    const unsigned char* a_base = (const unsigned char*)(g.A + (size_t)m0 * g.lda);
    const unsigned char* w_base = (const unsigned char*)(g.W + (size_t)n0 * g.ldw);
    uint32_t a_voff[8], w_voff[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int row = wave * 64 + r * 8 + (lane >> 3);
        const int chunk = ((lane & 7) ^ ((row >> 1) & 7)) << 4;
        int ar = row; ar = m0 + ar < g.M ? ar : g.M - 1 - m0;
        int wr = row; wr = n0 + wr < g.N ? wr : g.N - 1 - n0;
        a_voff[r] = (uint32_t)ar * (uint32_t)g.lda * (uint32_t)sizeof(T) + chunk;
        w_voff[r] = (uint32_t)wr * (uint32_t)g.ldw * (uint32_t)sizeof(T) + chunk;
    }
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, (short)0, 0x7fffffff, G4R_RSRC_WORD3);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)w_base, (short)0, 0x7fffffff, G4R_RSRC_WORD3);
    unsigned char* const my_rows = smem + wave * 64 * GEMM_ROW_BYTES;
    auto dma_a = [&](int kt, int r) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_ptr_t)(my_rows + (kt & 1) * G256_STAGE_BYTES + r * 8 * GEMM_ROW_BYTES), 16,
                                                 a_voff[r], kt * GEMM_ROW_BYTES, 0, 0);
    };
    auto dma_w = [&](int kt, int r) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(my_rows + (kt & 1) * G256_STAGE_BYTES + G256_OPERAND_BYTES + r * 8 * GEMM_ROW_BYTES), 16,
                                                 w_voff[r], kt * GEMM_ROW_BYTES, 0, 0);
    };

    f32x4 acc[8][8];                 // 128x128 per wave as 8x8 tiles of 16x16
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    // fragment of a 16x16x32 MFMA: lane l holds row (l & 15), K elements (l >> 4)*8 .. +7 of a 32-wide K block,
    // i.e. 16-byte chunk kb*4 + (l >> 4) of the 128-byte row; 16-row steps leave the swizzle unchanged
    int a_off[2], w_off[2];
    {
        const int l15 = lane & 15, kq = lane >> 4;
        const int swz = (l15 >> 1) & 7;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int c = ((kb * 4 + kq) ^ swz) << 4;
            a_off[kb] = (wm * 128 + l15) * GEMM_ROW_BYTES + c;
            w_off[kb] = G256_OPERAND_BYTES + (wn * 128 + l15) * GEMM_ROW_BYTES + c;
        }
    }
    u32x4 fa[2][8], fw[2][8];
    auto read_a = [&](int stage, int kb, int i) {
        fa[kb][i] = *(const u32x4*)(smem + stage * G256_STAGE_BYTES + a_off[kb] + i * 16 * GEMM_ROW_BYTES);
    };
    auto read_w = [&](int stage, int kb, int j) {
        fw[kb][j] = *(const u32x4*)(smem + stage * G256_STAGE_BYTES + w_off[kb] + j * 16 * GEMM_ROW_BYTES);
    };

    const int nk = g.K / BK;
    // ---- prologue: steps 0 and 1 requested, step 0 landed, its block-0 fragments read
#pragma unroll
    for (int r = 0; r < 8; ++r) dma_w(0, r);
#pragma unroll
    for (int r = 0; r < 8; ++r) dma_a(0, r);
    if (nk > 1) {
#pragma unroll
        for (int r = 0; r < 8; ++r) dma_w(1, r);
#pragma unroll
        for (int r = 0; r < 8; ++r) dma_a(1, r);
        __builtin_amdgcn_s_waitcnt(g4d_wait_vm(16));
    } else {
        __builtin_amdgcn_s_waitcnt(g4d_wait_vm(0));
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 8; ++j) read_w(0, 0, j);
#pragma unroll
    for (int i = 0; i < 8; ++i) read_a(0, 0, i);

    // more: step kt+1 exists (its block-0 fragments are read here); more2: step kt+2 exists (requested here);
    // WV: the wave this copy of the loop belongs to (its request slots).  One MFMA per scheduling region.
    auto step = [&](int kt, auto more_c, auto more2_c, auto wave_c) {
        constexpr bool more = decltype(more_c)::value, more2 = decltype(more2_c)::value;
        constexpr int WV = decltype(wave_c)::value;
        const int cur = kt & 1;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int p = kb * 64 + i * 8 + j;
            if (p == 36 && more2) {         // this wave has every fragment of stage cur in registers
                __builtin_amdgcn_s_waitcnt(G4R_WAIT_LGKM0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            if (p == 102 && more) {         // the 16 requests of this step may be in flight, those of the previous one not
                __builtin_amdgcn_s_waitcnt(g4d_wait_vm(more2 ? 16 : 0));
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            mfma16_agpr<T>(acc[i][j], fa[kb][i], fw[kb][j]);
            if (p < 16 && (p & 1) == 0) read_w(cur, 1, p >> 1);
            if (p >= 16 && p <= 30 && (p & 1) == 0) read_a(cur, 1, (p - 16) >> 1);
            if (more2 && p >= 38 && p < 70 && WV == ((p - 38) & 3)) dma_w(kt + 2, (p - 38) >> 2);
            if (more2 && p >= 70 && p < 102 && WV == ((p - 70) & 3)) dma_a(kt + 2, (p - 70) >> 2);
            if (more && p >= 103 && p <= 110) read_w(cur ^ 1, 0, p - 103);
            if (more && p >= 111 && p <= 125 && (p & 1) == 1) read_a(cur ^ 1, 0, (p - 111) >> 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    typedef std::integral_constant<bool, true> yes_t;
    typedef std::integral_constant<bool, false> no_t;
    auto k_loop = [&](auto wave_c) {
        int kt = 0;
        for (; kt + 2 < nk; ++kt) step(kt, yes_t{}, yes_t{}, wave_c);
        if (kt + 1 < nk) { step(kt, yes_t{}, no_t{}, wave_c); ++kt; }
        step(kt, no_t{}, no_t{}, wave_c);
    };
    if (wave == 0) k_loop(std::integral_constant<int, 0>{});
    else if (wave == 1) k_loop(std::integral_constant<int, 1>{});
    else if (wave == 2) k_loop(std::integral_constant<int, 2>{});
    else k_loop(std::integral_constant<int, 3>{});

    // ---- epilogue: each wave stages its 128x128 quadrant through a private 32 KiB LDS region
    // (64 rows x 128 fp32), two passes, drained by EpiDrain (gemm256.hip.h).
    asm volatile("s_nop 15\n\ts_nop 15");     // last MFMA (8 passes) -> first accumulator read
    __syncthreads();
    float* region = (float*)(smem + wave * 32768);
    typedef EpiDrain<T, ACT, RES, 64, 128, true, false> Drain;
    // the lane-derived indices of the epilogue are recomputed from an opaque copy of the thread id: kept alive
    // across the K loop (the compiler shares them with the prologue's) they are what no longer fits in 256 VGPRs
    int tid_e = tid;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63;
    const int l15 = lane_e & 15, kq = lane_e >> 4;
    const int gcol = n0 + wn * 128 + (lane_e % Drain::LPR) * 8;
    const bool col_ok = gcol < g.N;
    float4 bias8[2], sc8[2], sh8[2];
    Drain::load_cols(g.epi, gcol, col_ok, bias8, sc8, sh8);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        float4 oa[Drain::NIT], ob[Drain::NIT];
        const int row0 = m0 + wm * 128 + p * 64;
        Drain::load_res(g, row0, gcol, col_ok, lane_e, oa, ob);
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    region[(i4 * 16 + kq * 4 + r) * 128 + j * 16 + l15] = acc[4 * p + i4][j][r];
        if (RES || p == 0) __builtin_amdgcn_s_waitcnt(GEMM_WAIT_VMCNT0);
        Drain::drain(g, region, row0, gcol, col_ok, lane_e, bias8, sc8, sh8, oa, ob);
    }
}

template <typename T, int ACT, bool RES>
inline hipError_t launch_gemm4d_inst(const GemmArgs<T>& g, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm4d_tn_kernel<T, ACT, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int tiles_m = (g.M + G256_BM - 1) / G256_BM;
    const int tiles_n = (g.N + G256_BN - 1) / G256_BN;
    if (tiles_m <= 0 || tiles_n <= 0) return hipSuccess;
    hipLaunchKernelGGL((gemm4d_tn_kernel<T, ACT, RES>), dim3(tiles_m * tiles_n), dim3(256), G256_LDS_BYTES, stream, g);
    return hipGetLastError();
}

template <typename T, int ACT>
inline hipError_t launch_gemm4d_act(const GemmArgs<T>& g, hipStream_t stream) {
    return g.epi.residual ? launch_gemm4d_inst<T, ACT, true>(g, stream) : launch_gemm4d_inst<T, ACT, false>(g, stream);
}

template <typename T>
inline hipError_t launch_gemm4d(const GemmArgs<T>& g, hipStream_t stream) {
    switch (g.epi.act) {
        case ACT_GELU_TANH: return launch_gemm4d_act<T, ACT_GELU_TANH>(g, stream);
        case ACT_GELU_ERF: return launch_gemm4d_act<T, ACT_GELU_ERF>(g, stream);
        default: return launch_gemm4d_act<T, ACT_NONE>(g, stream);
    }
}

}  // namespace zett
