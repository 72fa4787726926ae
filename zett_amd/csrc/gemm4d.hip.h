// gemm4d.hip.h — tile variant 7: 256x256 tile, FOUR waves (one per SIMD, 128x128 of the tile each), both operands
// streamed HBM/L2 -> LDS by buffer_load_dwordx4 ... lds (no VGPR round trip, no ds_write pass), on
// v_mfma_f32_16x16x32_{bf16,f16} with the 256 accumulator registers pinned to AGPRs.  The tile of every 16-bit
// launch with K >= 2048: 4-8 % faster than the register-staged eight-wave kernels on the shapes of the benchmark step,
// residual or not (tools/gemm_bench), 5 % slower at K = 1024.  Same geometry as the hipBLASLt kernel the yardstick
// runs; the experiments that led here (ablations, K-start staggering, schedules with 3-4 barriers) are in
// gemm4dx.hip.h on the `experiments` branch.
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
// LDS image and swizzle of gemm_tile.hip.h: per stage and operand 256 rows x 128 B, 16-byte chunks
// XOR-swizzled by (row>>1)&7, applied to each lane's SOURCE address and undone on the ds_read_b128 side.
// The K reduction order per accumulator is that of every other tile variant.
//
// Epilogue (r2).  A per-tile timeline (trace4d.hip, `experiments` branch) showed 8 / 17 / 19 us of epilogue (plain bf16
// / erf-GELU / fp32 residual) behind an 80-88 us K loop at K = 4096, none of it memory latency: the generic drain of
// gemm_tile.hip.h spends its time on run-time epilogue flags (select chains, exec-mask branches), on a dependent
// packed-FMA chain per pair of GELU values, and on 16-byte accesses with 16-byte holes for fp32 rows (half the
// per-CU rate alone, a tenth under load: store_bench.hip, `experiments` branch).  The three epilogues that carry the
// benchmark step are therefore compiled as their own instantiations (EPI below): everything about the output is a
// template parameter, fp32 rows and residual rows are read and written as whole 512-byte row segments (four
// columns per lane), the 16-bit output as 256-byte segments (eight columns per lane), the staged rows are padded
// by four floats so both read patterns are conflict-free without per-lane reordering, the residual rows of BOTH
// passes are requested before the first store leaves (loads and stores share one in-order counter: a load issued
// behind a store cannot be waited for without waiting for the store), the GELUs are evaluated stage by stage over
// all values of a lane so that the chains interleave, and the last barrier moves into the last K step so that a
// wave starts staging as soon as its own MFMAs are done.  Same arithmetic per element (epi_value's operations in
// epi_value's order): identical bits, checked against the generic drain (gemm_variant 8) by the tile-variant test.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <type_traits>

#include "gemm_tile.hip.h"

namespace zett {

// Ablation hook for tools/gemm4d_ablate.hip ONLY (the library never defines it: level 5 = the product kernel, every condition
// below folds away).  Levels, cumulative, on REAL operands (the LDS stages keep the random tiles the prologue loaded, so the
// MFMAs toggle real data at every level): 1 = the MFMAs of the K loop alone, 2 = + the fragment reads (ds_read_b128), 3 = + the
// LDS-DMA requests, 4 = + the waits and barriers (= the whole K loop), 5 = + the epilogue.
#ifndef G4D_ABLATE
#define G4D_ABLATE 5
#endif
constexpr int G4D_ABL = G4D_ABLATE;
// Second hook of the same harness (r6): -DG4D_MFMA32 runs the K loop of the full tile on v_mfma_f32_32x32x16_{f16,bf16}
// (a wave's 128x128 as 4 x 4 blocks of 32x32; same LDS image, same request / wait / barrier slots, half as many MFMAs of twice the
// work) — levels 1-4 only (no epilogue: the accumulator layout differs).  The library never defines it.
#ifdef G4D_MFMA32
constexpr bool G4D_M32 = true;
static_assert(G4D_ABLATE < 5, "the 32x32x16 K loop exists for the ablation levels without an epilogue");
#else
constexpr bool G4D_M32 = false;
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));

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

template <typename T> __device__ __forceinline__ void mfma32_agpr(f32x16& c, const u32x4& a, const u32x4& b);
template <> __device__ __forceinline__ void mfma32_agpr<bf16_t>(f32x16& c, const u32x4& a, const u32x4& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
template <> __device__ __forceinline__ void mfma32_agpr<f16_t>(f32x16& c, const u32x4& a, const u32x4& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

// value of lane (l ^ K) within each group of 32 lanes (ds_swizzle bit mode: no LDS access, K < 32)
template <int K> __device__ __forceinline__ float g4d_xor_lane(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (K << 10) | 0x1f));
}


// EPI: which epilogue the instantiation carries.
//   GENERIC    EpiDrain of gemm_tile.hip.h: every combination of outputs, decided at run time
//   LO         only the 16-bit output (out_lo), bias + ACT            (QKV, FFN up, ProjectorBlock dense1)
//   F32        only the fp32 output (out_f32), bias + ACT + residual  (attention output, FFN down, dense2, heads)
//   F32_SCALE  as F32 with the Rescaler (scale, shift)                (output heads)
//   BOTH       fp32 AND 16-bit output of the same values, bias only   (input_projection.0: residual + operand of the ProjectorBlock)
//   F32_LN     as F32 with a residual, plus the 16-bit copy of the fp32 rows (out_lo) and per-row partial statistics over
//              the wave's 128 columns (stats_part): the producer half of the LayerNorm fold
//   LO_FOLD    as LO on an un-normalised operand: acc <- rstd_row * (acc - mean_row * fold_c[col]) in front of the bias
//   LO_LN      the producer of the output heads' ProjectorBlock LayerNorm (dense2: tanh-GELU + plain fp32 residual): ONLY the
//              16-bit copy of the sum and the partial statistics leave — nothing reads the fp32 sum again
//   F32_SCALE_FOLD   its consumer: the final Linear with the Rescaler, on the un-normalised operand (as LO_FOLD, fp32 output)
//   LN16       (r4) the encoder's producer on a 16-BIT RESIDUAL STREAM: the residual rows come from the 16-bit copy of the hidden
//              state (residual_lo; LayerNorm'd on the fly, or as stored; indexed or not), and only the 16-bit copy of the sum and
//              the partial statistics leave — 4 instead of 10 bytes per element, eight columns per lane on both sides (one
//              16-byte load and one 16-byte store per eight values where F32_LN issues a 16-byte load, a 16-byte and an
//              8-byte store per four).
//              Round 3 had tried the 16-bit stream on F32_LN's four-column lane mapping (8-byte loads: slower, NOTEBOOK R3.3).
enum { G4D_EPI_GENERIC = 0, G4D_EPI_LO = 1, G4D_EPI_F32 = 2, G4D_EPI_F32_SCALE = 3, G4D_EPI_BOTH = 4, G4D_EPI_F32_LN = 5, G4D_EPI_LO_FOLD = 6,
       G4D_EPI_LO_LN = 7, G4D_EPI_F32_SCALE_FOLD = 8, G4D_EPI_LN16 = 9 };
constexpr int G4D_EPI_STRIDE = 132;                                   // floats per staged row: 128 columns + 4 of padding
constexpr int G4D_EPI_REGION = 64 * G4D_EPI_STRIDE * 4;               // bytes per wave
constexpr int G4D_LDS_BYTES = 4 * G4D_EPI_REGION;                     // 132 KiB (the K loop uses the first 128)
static_assert(G4D_LDS_BYTES >= G256_LDS_BYTES && G4D_LDS_BYTES <= 160 * 1024, "LDS budget");

// Geometry of the two tiles.  NOTE for whoever edits this kernel: local arrays whose BOUND depends on HALF (a_voff[AREQ],
// acc[MI][8], fa[2][MI]) made clang's HOST pass silently drop the kernel's stub (no diagnostic; the objects then carry every
// gemm4d kernel as an undefined symbol and the library does not load) — the arrays therefore keep the full tile's bounds and the
// HALF tile uses their first half.  `hipcc --cuda-host-only -S -emit-llvm` + grep "define.*__device_stub__" shows it in a second.
template <bool HALF> struct G4dGeom {
    static constexpr int BM = HALF ? 128 : G256_BM;      // rows of the tile
    static constexpr int WROWS = BM / 2;                 // rows of a wave's quadrant
    static constexpr int MI = WROWS / 16;                // 16-row MFMA blocks per wave
    static constexpr int AREQ = BM / 32;                 // LDS-DMA requests per wave, K step and A image (8 rows each)
    static constexpr int NREQ = AREQ + 8;                // ... and per K step in all
    static constexpr int NPASS = WROWS / 64;             // epilogue passes of 64 rows
    // LDS stages of the K loop.  The full tile double-buffers (2 x 64 KiB); the HALF tile's K step is half as long (64 MFMAs), so
    // a request issued one step ahead has half the time to land — measured: the half tiles of a launch's last round took 0.88 of
    // a full round instead of ~0.55 — and its stage is 48 KiB (A 16 + W 32): it runs THREE stages, requests two steps ahead
    static constexpr int STAGES = HALF ? 3 : 2;
    static constexpr int A_BYTES = BM * GEMM_ROW_BYTES;                     // A image of a stage (W image behind it)
    static constexpr int STAGE_BYTES = A_BYTES + G256_OPERAND_BYTES;
    static constexpr int LDS_BYTES = STAGES * STAGE_BYTES;
};

// HALF (r5): the same kernel on a 128x256 tile — four waves of 64x128 each (128 accumulators), covering rows [g.row0, g.M) only.
// It exists for ONE purpose: the partly filled last round of a launch.  A launch whose 256x256 tiles fill R whole rounds of the
// 256 CUs and a fraction of one more is cut by the launcher (gemm4d_row_split) into full tiles on rows [0, row0) — whole rounds —
// and HALF tiles on the rest: twice as many workgroups of half the work each, so the last round's tiles spread over the idle CUs
// (M = 9 682 at N = 4096, a rank's shard at 8 GPUs: 608 tiles = 2.375 rounds -> 512 tiles + 192 half tiles).  A HALF tile's
// K step is 64 MFMAs against 12 LDS-DMA requests and 12 + 12 fragment reads (the full tile: 128 against 16 and 16 + 16), so its
// K loop is slower per FLOP (NOTEBOOK R3.1 measured 28 % for two half tiles per CU) — it only ever replaces idle CUs.  Same K
// order per accumulator, same epilogue arithmetic: identical bits (tests/test_invariants_gpu.py).
template <typename T, int ACT = ACT_NONE, bool RES = false, int EPI = G4D_EPI_GENERIC, bool HALF = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm4d_tn_kernel(GemmArgs<T> g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BK = GEMM_ROW_BYTES / (int)sizeof(T);
    typedef G4dGeom<HALF> GEO;          // (static members, not constexpr locals: see G4dGeom)
    static_assert(!HALF || EPI != G4D_EPI_GENERIC, "the half tile carries the streamlined epilogues only");

    const int tiles_m = (g.M - g.row0 + GEO::BM - 1) / GEO::BM;
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
    const int GROUP_N = g.group > 0 ? g.group : 4;
    int tm, tn;
    if (g.tile_order == 0) {
        const int group_size = GROUP_N * tiles_m;
        const int first_n = (wg / group_size) * GROUP_N;
        const int gn = (tiles_n - first_n) < GROUP_N ? (tiles_n - first_n) : GROUP_N;
        tn = first_n + (wg % group_size) % gn;
        tm = (wg % group_size) / gn;
    } else {                      // zett_set_option("gemm_tile_order", 1): ZETT_GROUP_M row tiles first, as gemm8r / gemm8x
        const int GROUP_M = g.group > 0 ? g.group : ZETT_GROUP_M;
        const int group_size = GROUP_M * tiles_n;
        const int first_m = (wg / group_size) * GROUP_M;
        const int gm = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
        tm = first_m + (wg % group_size) % gm;
        tn = (wg % group_size) / gm;
    }
    const int m0 = g.row0 + tm * GEO::BM, n0 = tn * G256_BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // 0..3
    const int wm = wave >> 1, wn = wave & 1;

    // request r (0..7) of a wave moves rows wave*64 + r*8 + lane/8 of an operand; the lane's LDS slot is
    // chunk lane%8 of its row, which holds source chunk (lane%8) ^ swz(row).  Rows past the edge are clamped.
    const unsigned char* a_base = (const unsigned char*)(g.A + (size_t)m0 * g.lda);
    const unsigned char* w_base = (const unsigned char*)(g.W + (size_t)n0 * g.ldw);
    // (HALF: the A image is 128 rows — request r (0..3) of a wave moves rows wave*32 + r*8 + lane/8)
    uint32_t a_voff[8], w_voff[8];          // (HALF uses a_voff[0..3]; arrays whose bound depends on HALF make clang drop the host stub, see G4dGeom)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int row = wave * 64 + r * 8 + (lane >> 3);
        const int chunk = ((lane & 7) ^ ((row >> 1) & 7)) << 4;
        int wr = row; wr = n0 + wr < g.N ? wr : g.N - 1 - n0;
        w_voff[r] = (uint32_t)wr * (uint32_t)g.ldw * (uint32_t)sizeof(T) + chunk;
    }
#pragma unroll
    for (int r = 0; r < GEO::AREQ; ++r) {
        const int row = wave * (GEO::AREQ * 8) + r * 8 + (lane >> 3);
        const int chunk = ((lane & 7) ^ ((row >> 1) & 7)) << 4;
        int ar = row; ar = m0 + ar < g.M ? ar : g.M - 1 - m0;
        a_voff[r] = (uint32_t)ar * (uint32_t)g.lda * (uint32_t)sizeof(T) + chunk;
    }
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, (short)0, 0x7fffffff, G4R_RSRC_WORD3);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)w_base, (short)0, 0x7fffffff, G4R_RSRC_WORD3);
    unsigned char* const my_rows = smem + wave * 64 * GEMM_ROW_BYTES;
    unsigned char* const my_a_rows = smem + wave * (GEO::AREQ * 8) * GEMM_ROW_BYTES;
    auto dma_a = [&](int stage, int kt, int r) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_ptr_t)(my_a_rows + stage * GEO::STAGE_BYTES + r * 8 * GEMM_ROW_BYTES), 16,
                                                 a_voff[r], kt * GEMM_ROW_BYTES, 0, 0);
    };
    auto dma_w = [&](int stage, int kt, int r) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(my_rows + stage * GEO::STAGE_BYTES + GEO::A_BYTES + r * 8 * GEMM_ROW_BYTES), 16,
                                                 w_voff[r], kt * GEMM_ROW_BYTES, 0, 0);
    };

    constexpr bool M32 = G4D_M32 && !HALF;
    f32x4 acc[8][8];                // 128x128 (HALF: 64x128) per wave as GEO::MI x 8 tiles of 16x16
    f32x16 acc32[4][4];             // (ablation, G4D_MFMA32: the same 128x128 as 4 x 4 tiles of 32x32; constant bounds, see G4dGeom; unused -> no registers)
    if constexpr (M32) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc32[i][j][r] = 0.f;
    } else {
#pragma unroll
    for (int i = 0; i < GEO::MI; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    }

    // fragment of a 16x16x32 MFMA: lane l holds row (l & 15), K elements (l >> 4)*8 .. +7 of a 32-wide K block,
    // i.e. 16-byte chunk kb*4 + (l >> 4) of the 128-byte row; 16-row steps leave the swizzle unchanged
    int a_off[2], w_off[2];
    {
        const int l15 = lane & 15, kq = lane >> 4;
        const int swz = (l15 >> 1) & 7;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int c = ((kb * 4 + kq) ^ swz) << 4;
            a_off[kb] = (wm * GEO::WROWS + l15) * GEMM_ROW_BYTES + c;
            w_off[kb] = GEO::A_BYTES + (wn * 128 + l15) * GEMM_ROW_BYTES + c;
        }
    }
    // (G4D_MFMA32) fragment of a 32x32x16 MFMA: lane l holds row (l & 31), K elements (l >> 5)*8 .. +7 of a 16-wide K block, i.e.
    // 16-byte chunk k16*2 + (l >> 5) of the row; 32-row steps leave the swizzle unchanged.  Read q (0..7) of a 64-wide K block kb:
    // K block k16 = kb*2 + q/4, row block q%4 — the same eight reads per operand and K block as the 16x16x32 loop issues.
    int a_off32[4], w_off32[4];
    if constexpr (M32) {
        const int l31 = lane & 31, kh = lane >> 5;
        const int swz = (l31 >> 1) & 7;
#pragma unroll
        for (int k16 = 0; k16 < 4; ++k16) {
            const int c = ((k16 * 2 + kh) ^ swz) << 4;
            a_off32[k16] = (wm * GEO::WROWS + l31) * GEMM_ROW_BYTES + c;
            w_off32[k16] = GEO::A_BYTES + (wn * 128 + l31) * GEMM_ROW_BYTES + c;
        }
    }
    u32x4 fa[2][8], fw[2][8];
    auto read_a = [&](int stage, int kb, int i) {
        if constexpr (M32) fa[kb][i] = *(const u32x4*)(smem + stage * GEO::STAGE_BYTES + a_off32[kb * 2 + (i >> 2)] + (i & 3) * 32 * GEMM_ROW_BYTES);
        else fa[kb][i] = *(const u32x4*)(smem + stage * GEO::STAGE_BYTES + a_off[kb] + i * 16 * GEMM_ROW_BYTES);
    };
    auto read_w = [&](int stage, int kb, int j) {
        if constexpr (M32) fw[kb][j] = *(const u32x4*)(smem + stage * GEO::STAGE_BYTES + w_off32[kb * 2 + (j >> 2)] + (j & 3) * 32 * GEMM_ROW_BYTES);
        else fw[kb][j] = *(const u32x4*)(smem + stage * GEO::STAGE_BYTES + w_off[kb] + j * 16 * GEMM_ROW_BYTES);
    };

    const int nk = g.K / BK;
    // LN16 (16-bit residual stream): the residual rows of pass 0 — sixteen 16-byte loads per lane, 64 registers — are requested
    // from the LAST K step, one per four MFMAs in the request slots that step leaves empty, so that their latency runs under
    // the MFMAs instead of at the head of the epilogue (the fp32 residual's 128 registers do not exist beside the fragments).
    // Not for indexed residual rows (layer 0 with the pair lever: the row indices are fetched in the epilogue).
    constexpr bool LN16K = EPI == G4D_EPI_LN16;
    uint4 res16[LN16K ? 16 : 1];
    const bool res16_early = LN16K && g.epi.res_index == nullptr;
    // row t*4 + lane/16 of pass p of this wave, columns (lane%16)*8 .. +7; rows / columns past the edge are clamped (their values are never stored)
    auto res16_request = [&](int p, int t) __attribute__((always_inline)) {
        if constexpr (LN16K) {
            int grow = m0 + wm * GEO::WROWS + p * 64 + t * 4 + (lane >> 4);
            grow = grow < g.M ? grow : g.M - 1;
            int gc = n0 + wn * 128 + (lane & 15) * 8;
            gc = gc < g.N ? gc : 0;
            res16[t] = *(const uint4*)(g.epi.residual_lo + (size_t)grow * g.epi.ld_res_lo + gc);
        }
    };
    // ---- prologue: the first STAGES steps requested, step 0 landed, its block-0 fragments read
#pragma unroll
    for (int st = 0; st < GEO::STAGES; ++st) {
        if (st < nk) {
#pragma unroll
            for (int r = 0; r < 8; ++r) dma_w(st, st, r);
#pragma unroll
            for (int r = 0; r < GEO::AREQ; ++r) dma_a(st, st, r);
        }
    }
    {
        const int ahead = (nk < GEO::STAGES ? nk : GEO::STAGES) - 1;      // requested steps behind step 0
        if (ahead >= 2) __builtin_amdgcn_s_waitcnt(g4d_wait_vm(2 * GEO::NREQ));
        else if (ahead == 1) __builtin_amdgcn_s_waitcnt(g4d_wait_vm(GEO::NREQ));
        else __builtin_amdgcn_s_waitcnt(g4d_wait_vm(0));
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 8; ++j) read_w(0, 0, j);
#pragma unroll
    for (int i = 0; i < GEO::MI; ++i) read_a(0, 0, i);
    if constexpr (G4D_ABL < 2) {          // (ablation: no fragment reads in the loop — both K blocks' fragments are read once, here)
#pragma unroll
        for (int j = 0; j < 8; ++j) read_w(0, 1, j);
#pragma unroll
        for (int i = 0; i < GEO::MI; ++i) read_a(0, 1, i);
        __builtin_amdgcn_s_waitcnt(G4R_WAIT_LGKM0);
    }

    // more: step kt+1 exists (its block-0 fragments are read here); more2: step kt+2 exists; moreA: step kt+STAGES exists (requested
    // here, into the stage this step computes on: free once every wave has its block-1 fragments);
    // WV: the wave this copy of the loop belongs to (its request slots).  One MFMA per scheduling region.
    auto step = [&](int kt, int cur, auto more_c, auto more2_c, auto moreA_c, auto wave_c) {
        constexpr bool more = decltype(more_c)::value, more2 = decltype(more2_c)::value, moreA = decltype(moreA_c)::value;
        constexpr int WV = decltype(wave_c)::value;
        const int nxt = cur + 1 == GEO::STAGES ? 0 : cur + 1;
        // Slots of the K step (p = the MFMA a piece of work is issued behind).  Full tile: the header's schedule.  HALF tile: 64
        // MFMAs — block-1 fragments 0..22, barrier 26, the 12 requests of step t+3 at 27..50 (two waves per slot), vmcnt +
        // barrier 52, block-0 fragments of step t+1 at 53..63.
        constexpr int P_BAR1 = HALF ? 26 : 36, P_BAR2 = HALF ? 52 : 102;
        // requests that may still be in flight when step kt+1's must have landed: those of the later steps that exist
        constexpr int INFLIGHT = HALF ? ((more2 ? 1 : 0) + (moreA ? 1 : 0)) * GEO::NREQ : (moreA ? GEO::NREQ : 0);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < GEO::MI; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int p = kb * (GEO::MI * 8) + i * 8 + j;
            // (last step, streamlined epilogues: the barrier every wave passes after its last fragment read, so that
            //  staging the accumulators over the operand images needs no barrier behind the loop)
            if (G4D_ABL >= 4 && p == P_BAR1 && (moreA || (!more && EPI != G4D_EPI_GENERIC))) {         // this wave has every fragment of stage cur in registers
                __builtin_amdgcn_s_waitcnt(G4R_WAIT_LGKM0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            if (G4D_ABL >= 4 && p == P_BAR2 && more) {         // the requests of the later steps may be in flight, those of step kt+1 not
                __builtin_amdgcn_s_waitcnt(g4d_wait_vm(INFLIGHT));
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (M32) {
                // slot p of the 16x16x32 schedule; MFMA p/2 of the 32x32x16 loop is issued in front of the even slot: K block
                // k16 = kb*2 + h, row block i32, column block j32 (64 MFMAs of 16 passes where the product loop issues 128 of 8)
                if ((p & 1) == 0) {
                    const int q = (p >> 1) & 31, h = q >> 4, i32 = (q >> 2) & 3, j32 = q & 3;
                    mfma32_agpr<T>(acc32[i32][j32], fa[kb][h * 4 + i32], fw[kb][h * 4 + j32]);
                }
            } else
            mfma16_agpr<T>(acc[i][j], fa[kb][i], fw[kb][j]);
            constexpr bool RD = G4D_ABL >= 2, DMA = G4D_ABL >= 3;          // (ablation hook: always true in the library)
            if (RD && p < 16 && (p & 1) == 0) read_w(cur, 1, p >> 1);
            if constexpr (!HALF) {
                if (RD && p >= 16 && p <= 30 && (p & 1) == 0) read_a(cur, 1, (p - 16) >> 1);
                if (DMA && moreA && p >= 38 && p < 70 && WV == ((p - 38) & 3)) dma_w(cur, kt + 2, (p - 38) >> 2);
                if (DMA && moreA && p >= 70 && p < 102 && WV == ((p - 70) & 3)) dma_a(cur, kt + 2, (p - 70) >> 2);
                if (RD && more && p >= 103 && p <= 110) read_w(nxt, 0, p - 103);
                if (RD && more && p >= 111 && p <= 125 && (p & 1) == 1) read_a(nxt, 0, (p - 111) >> 1);
                if constexpr (LN16K) { if (!more && p >= 40 && p < 104 && (p & 3) == 0 && res16_early) res16_request(0, (p - 40) >> 2); }
            } else {
                if (p >= 16 && p <= 22 && (p & 1) == 0) read_a(cur, 1, (p - 16) >> 1);
                if (moreA && p >= 27 && p < 43 && (WV & 1) == ((p - 27) & 1)) dma_w(cur, kt + 3, (p - 27) >> 1);
                if (moreA && p >= 43 && p < 51 && (WV & 1) == ((p - 43) & 1)) dma_a(cur, kt + 3, (p - 43) >> 1);
                if (more && p >= 53 && p <= 60) read_w(nxt, 0, p - 53);
                if (more && p >= 60 && p <= 63) read_a(nxt, 0, p - 60);
                if constexpr (LN16K) { if (!more && p >= 24 && p < 56 && (p & 1) == 0 && res16_early) res16_request(0, (p - 24) >> 1); }
            }
            if (!M32 || (p & 1)) __builtin_amdgcn_sched_barrier(0);
        }
    };
    typedef std::integral_constant<bool, true> yes_t;
    typedef std::integral_constant<bool, false> no_t;
    auto k_loop = [&](auto wave_c) {
        int kt = 0, cur = 0;
        auto next = [&]() { ++kt; cur = cur + 1 == GEO::STAGES ? 0 : cur + 1; };
        for (; kt + GEO::STAGES < nk; next()) step(kt, cur, yes_t{}, yes_t{}, yes_t{}, wave_c);
        if constexpr (HALF) {
            if (kt + 2 < nk) { step(kt, cur, yes_t{}, yes_t{}, no_t{}, wave_c); next(); }
        }
        if (kt + 1 < nk) { step(kt, cur, yes_t{}, no_t{}, no_t{}, wave_c); next(); }
        step(kt, cur, no_t{}, no_t{}, no_t{}, wave_c);
    };
    if (wave == 0) k_loop(std::integral_constant<int, 0>{});
    else if (wave == 1) k_loop(std::integral_constant<int, 1>{});
    else if (wave == 2) k_loop(std::integral_constant<int, 2>{});
    else k_loop(std::integral_constant<int, 3>{});

    asm volatile("s_nop 15\n\ts_nop 15");     // last MFMA (8 passes) -> first accumulator read
    if constexpr (G4D_ABL < 5) {          // (ablation: no epilogue — one value per lane keeps the accumulators alive)
        float keep = 0.f;
        if constexpr (M32) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) keep += acc32[i][j][r];
        } else {
#pragma unroll
        for (int i = 0; i < GEO::MI; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) keep += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        }
        if (keep == 123.456f) g.epi.out_lo[threadIdx.x] = (T)0;
        return;
    }
    // the lane-derived indices of the epilogue are recomputed from an opaque copy of the thread id: kept alive
    // across the K loop (the compiler shares them with the prologue's) they are what no longer fits in 256 VGPRs
    int tid_e = tid;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63;
    const int l15 = lane_e & 15, kq = lane_e >> 4;
    if constexpr (EPI == G4D_EPI_GENERIC) {
        // ---- generic epilogue: each wave stages its 128x128 quadrant through a private 32 KiB LDS region
        // (64 rows x 128 fp32), two passes, drained by EpiDrain (gemm_tile.hip.h).
        __syncthreads();
        float* region = (float*)(smem + wave * 32768);
        typedef EpiDrain<T, ACT, RES, 64, 128, !RES, false> Drain;      // (no Rescaler behind a residual: the launcher refuses the pair)
        const int gcol = n0 + wn * 128 + (lane_e % Drain::LPR) * 8;
        const bool col_ok = gcol < g.N;
        float4 bias8[2], sc8[2], sh8[2];
        Drain::load_cols(g.epi, gcol, col_ok, bias8, sc8, sh8);
        float4 lng[2], lnb[2];
        Drain::load_ln_cols(g.epi, gcol, col_ok, lng, lnb);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float4 oa[Drain::NIT], ob[Drain::NIT];
            const int row0 = m0 + wm * 128 + p * 64;
            Drain::load_res(g, row0, gcol, col_ok, lane_e, oa, ob);
            float2 lnst;
            Drain::load_res_stats(g, row0, lane_e, lnst);
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        region[(i4 * 16 + kq * 4 + r) * 128 + j * 16 + l15] = acc[4 * p + i4][j][r];
            if (RES || p == 0) __builtin_amdgcn_s_waitcnt(GEMM_WAIT_VMCNT0);
            Drain::ln_res(g, lane_e, lng, lnb, lnst, oa, ob);
            Drain::drain(g, region, row0, gcol, col_ok, lane_e, bias8, sc8, sh8, oa, ob);
        }
    } else {
        // ---- streamlined epilogues (see the header).  No barrier here: the last K step carries it.
        constexpr bool LO = EPI == G4D_EPI_LO || EPI == G4D_EPI_LO_FOLD, SCALE = EPI == G4D_EPI_F32_SCALE || EPI == G4D_EPI_F32_SCALE_FOLD;
        constexpr bool LN16 = EPI == G4D_EPI_LN16;
        constexpr bool LNP = EPI == G4D_EPI_F32_LN || EPI == G4D_EPI_LO_LN || LN16, FOLD = EPI == G4D_EPI_LO_FOLD || EPI == G4D_EPI_F32_SCALE_FOLD;
        constexpr bool WF32 = EPI != G4D_EPI_LO_LN && !LN16;          // the fp32 output is written
        static_assert(!LNP || RES, "a LayerNorm producer is a residual epilogue");
        static_assert((EPI != G4D_EPI_F32_LN && !LN16) || ACT == ACT_NONE, "the encoder's producer has no activation");
        static_assert(!LN16 || sizeof(T) == 2, "16-bit residual stream");
        constexpr int CPL = (LO || LN16) ? 8 : 4;  // columns per lane
        constexpr int LPR = 128 / CPL;             // lanes per row
        constexpr int RPI = 64 / LPR;              // rows per wave instruction
        constexpr int NIT = 64 / RPI;              // instructions per pass
        static_assert(!(LO && RES), "the 16-bit-only epilogue has no residual");
        float* region = (float*)(smem + wave * G4D_EPI_REGION);
        const GemmEpilogue<T>& e = g.epi;
        // range guard (GemmEpilogue::range_flag): every value that leaves as a 16-bit operand of an f16 launch is compared
        // with the half range (one v_cmp per value, accumulated in a scalar mask); the launches that write the predicted
        // embeddings (range_final, wave-uniform) check their fp32 values for inf / NaN instead
        constexpr bool W16 = LO || EPI == G4D_EPI_BOTH || LNP;
        constexpr bool CHK16 = W16 && LoRange<T>::checked;
        const bool chk_final = !W16 && e.range_final != 0;
        bool bad = false;
        const int idx = lane_e % LPR, rsub = lane_e / LPR;
        const int gcol = n0 + wn * 128 + idx * CPL;
        const bool col_ok = gcol < g.N;
        const int grow0 = m0 + wm * GEO::WROWS + rsub;    // the lane's row in instruction t of pass p: grow0 + p*64 + t*RPI
        float bias[CPL], sc[CPL], sh[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) { bias[c] = 0.f; sc[c] = 1.f; sh[c] = 0.f; }
        if (col_ok) {
#pragma unroll
            for (int c4 = 0; c4 < CPL; c4 += 4) {
                if (e.bias) { const float4 b = *(const float4*)(e.bias + gcol + c4); bias[c4] = b.x; bias[c4 + 1] = b.y; bias[c4 + 2] = b.z; bias[c4 + 3] = b.w; }
                if (SCALE) {
                    const float4 a = *(const float4*)(e.scale + gcol + c4), b = *(const float4*)(e.shift + gcol + c4);
                    sc[c4] = a.x; sc[c4 + 1] = a.y; sc[c4 + 2] = a.z; sc[c4 + 3] = a.w;
                    sh[c4] = b.x; sh[c4 + 1] = b.y; sh[c4 + 2] = b.z; sh[c4 + 3] = b.w;
                }
            }
        }
        // LayerNorm'd residual (GemmEpilogue::res_stats): lane l holds the statistics of row l of each pass (two
        // 8-byte loads per lane for the whole tile); instruction t of the drain covers rows 2t and 2t+1 of the pass and
        // fetches theirs with v_readlane.  gamma / beta of the lane's four columns sit beside the bias.
        // (no activation in front of a LayerNorm'd residual: gemm4d_epi_mode sends anything else to the generic drain)
        const bool res_ln = RES && ACT == ACT_NONE && e.res_stats != nullptr;
        float lg[CPL], lb[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) { lg[c] = 1.f; lb[c] = 0.f; }
        float2 pst[2] = {make_float2(0.f, 1.f), make_float2(0.f, 1.f)};
        if constexpr (RES && ACT == ACT_NONE) {
            if (res_ln) {
                if (col_ok) {
#pragma unroll
                    for (int c4 = 0; c4 < CPL; c4 += 4) {
                        const float4 a = *(const float4*)(e.res_gamma + gcol + c4), b = *(const float4*)(e.res_beta + gcol + c4);
                        lg[c4] = a.x; lg[c4 + 1] = a.y; lg[c4 + 2] = a.z; lg[c4 + 3] = a.w; lb[c4] = b.x; lb[c4 + 1] = b.y; lb[c4 + 2] = b.z; lb[c4 + 3] = b.w;
                    }
                }
#pragma unroll
                for (int p = 0; p < GEO::NPASS; ++p) {
                    int srow = m0 + wm * GEO::WROWS + p * 64 + lane_e; srow = srow < g.M ? srow : g.M - 1;
                    if (e.res_index) srow = e.res_index[srow];
                    pst[p] = *(const float2*)(e.res_stats + 2 * (size_t)srow);
                }
            }
        }
        // indexed residual rows (GemmEpilogue::res_index): lane l holds the index of row l of each pass, fetched per
        // instruction with v_readlane like the statistics
        int rix[2] = {0, 0};
        const bool res_ix = RES && e.res_index != nullptr;
        if constexpr (RES) {
            if (res_ix) {
#pragma unroll
                for (int p = 0; p < GEO::NPASS; ++p) {
                    int srow = m0 + wm * GEO::WROWS + p * 64 + lane_e; srow = srow < g.M ? srow : g.M - 1;
                    rix[p] = e.res_index[srow];
                }
            }
        }
        // LayerNorm fold, consumer side: the row sums of the folded weight for the lane's columns, and (mean, rstd) of row l
        // of each pass in lane l (fetched per instruction with v_readlane)
        float fc[FOLD ? CPL : 1];
        float2 fst[2] = {make_float2(0.f, 1.f), make_float2(0.f, 1.f)};
        if constexpr (FOLD) {
#pragma unroll
            for (int c = 0; c < CPL; ++c) fc[c] = 0.f;
            if (col_ok) {
#pragma unroll
                for (int c4 = 0; c4 < CPL; c4 += 4) {
                    const float4 a = *(const float4*)(e.fold_c + gcol + c4);
                    fc[c4] = a.x; fc[c4 + 1] = a.y; fc[c4 + 2] = a.z; fc[c4 + 3] = a.w;
                }
            }
#pragma unroll
            for (int p = 0; p < GEO::NPASS; ++p) {
                int srow = m0 + wm * GEO::WROWS + p * 64 + lane_e; srow = srow < g.M ? srow : g.M - 1;
                fst[p] = *(const float2*)(e.fold_stats + 2 * (size_t)srow);
            }
        }
        // residual rows of both passes: requested before any store leaves (pass 1's right after pass 0 is staged, by
        // which time the accumulators of pass 0 have left their registers)
        // (LNP: ONE buffer — pass 1's rows are requested after pass 0 has drained; the row statistics and the second
        //  output need the registers that both passes' rows would take, and a spill costs more than that wait)
        constexpr int RB = LNP ? 1 : 2;
        float4 res[LN16 ? 1 : RB][(RES && !LN16) ? NIT : 1];
        // (LN16: res16[] above — eight 16-bit residual values per lane and instruction, 64 registers per pass; both passes at
        //  once — 128 — was tried first and spills: 456 bytes of scratch per lane)
        auto load_res16 = [&](int p, int ta, int tb) {      // instructions [ta, tb) of pass p
            if constexpr (LN16) {
#pragma unroll
                for (int t = ta; t < tb; ++t) {
                    if (res_ix) {
                        const int grow = grow0 + p * 64 + t * RPI;
                        const size_t rrow = (size_t)__shfl(rix[p], t * RPI + rsub, 64);       // row t*RPI + rsub of the pass
                        res16[t] = (grow < g.M && col_ok) ? *(const uint4*)(e.residual_lo + rrow * e.ld_res_lo + gcol) : make_uint4(0u, 0u, 0u, 0u);
                    } else {
                        res16_request(p, t);
                    }
                }
            }
        };
        auto load_res = [&](int p) {
            if constexpr (LN16) {
                load_res16(p, 0, NIT);
            } else if constexpr (RES) {
#pragma unroll
                for (int t = 0; t < NIT; ++t) {
                    const int grow = grow0 + p * 64 + t * RPI;
                    size_t rrow = (size_t)grow;
                    if (res_ix) {               // rows t*RPI (rsub = 0) and t*RPI + 1 (rsub = 1) of the pass
                        const int i0 = __builtin_amdgcn_readlane(rix[p], t * RPI), i1 = __builtin_amdgcn_readlane(rix[p], t * RPI + 1);
                        rrow = (size_t)(rsub ? i1 : i0);
                    }
                    res[p % RB][t] = (grow < g.M && col_ok) ? *(const float4*)(e.residual + rrow * e.ld_res + gcol) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        };
        auto stage = [&](int p) {
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        region[(i4 * 16 + kq * 4 + r) * G4D_EPI_STRIDE + j * 16 + l15] = acc[4 * p + i4][j][r];
            if constexpr (FOLD)      // row l's statistics into the four floats of padding behind its 128 staged columns
                *(float2*)(region + lane_e * G4D_EPI_STRIDE + 128) = fst[p];
            if constexpr (LN16)      // the same place for the statistics of row l's LayerNorm'd residual
                *(float2*)(region + lane_e * G4D_EPI_STRIDE + 128) = pst[p];
        };
        // FULL: all 64 rows and all 128 columns of the pass lie inside the matrix (wave-uniform): no predicate anywhere.
        // fp32 rows behind a residual go out non-temporal: they are streamed once to the LayerNorm kernel and stay out
        // of the L2 the operand panels live in (+3..6 % on those launches, tools/gemm_bench EPI=2).
        auto drain = [&](int p, auto full_c) {
            constexpr bool FULL = decltype(full_c)::value;
            // GROUP instructions at a time: their LDS reads, then their arithmetic stage by stage, then their stores
            constexpr int GROUP = (LO || RES) ? 4 : 8;      // 32 values per lane and group (16 beside the 256 residual registers)
            constexpr int NV = GROUP * CPL;
            float acc_s = 0.f, acc_q = 0.f;        // LNP: (sum, sum of squares) of the row this lane ends up holding
#pragma unroll
            for (int t0 = 0; t0 < NIT; t0 += GROUP) {
                float v[NV], bb[NV], rr[NV], ss[NV], hh[NV];
#pragma unroll
                for (int u = 0; u < GROUP; ++u) {
                    const float* src = region + ((t0 + u) * RPI + rsub) * G4D_EPI_STRIDE + idx * CPL;
#pragma unroll
                    for (int c4 = 0; c4 < CPL; c4 += 4) {
                        const float4 x = *(const float4*)(src + c4);
                        v[u * CPL + c4] = x.x; v[u * CPL + c4 + 1] = x.y; v[u * CPL + c4 + 2] = x.z; v[u * CPL + c4 + 3] = x.w;
                    }
#pragma unroll
                    for (int c = 0; c < CPL; ++c) { bb[u * CPL + c] = bias[c]; ss[u * CPL + c] = sc[c]; hh[u * CPL + c] = sh[c]; rr[u * CPL + c] = 0.f; }
                    if constexpr (LN16) {
                        const uint4 x = res16[t0 + u];
                        unpack2_lo<T>(x.x, rr[u * 8], rr[u * 8 + 1]); unpack2_lo<T>(x.y, rr[u * 8 + 2], rr[u * 8 + 3]);
                        unpack2_lo<T>(x.z, rr[u * 8 + 4], rr[u * 8 + 5]); unpack2_lo<T>(x.w, rr[u * 8 + 6], rr[u * 8 + 7]);
                        if (res_ln) {
                            const float2 st = *(const float2*)(region + ((t0 + u) * RPI + rsub) * G4D_EPI_STRIDE + 128);
#pragma unroll
                            for (int c = 0; c < 8; ++c) rr[u * 8 + c] = ln_affine(rr[u * 8 + c], st.x, st.y, lg[c], lb[c]);
                        }
                    } else if constexpr (RES) {
                        const float4 x = res[p % RB][t0 + u];
                        rr[u * CPL] = x.x; rr[u * CPL + 1] = x.y; rr[u * CPL + 2] = x.z; rr[u * CPL + 3] = x.w;
                        if (ACT == ACT_NONE && res_ln) {
                            const int r0 = (t0 + u) * RPI;          // rows r0 (rsub = 0) and r0 + 1 (rsub = 1) of the pass
                            const float mean0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pst[p].x), r0));
                            const float mean1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pst[p].x), r0 + 1));
                            const float rstd0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pst[p].y), r0));
                            const float rstd1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pst[p].y), r0 + 1));
                            const float mean = rsub ? mean1 : mean0, rstd = rsub ? rstd1 : rstd0;
#pragma unroll
                            for (int c = 0; c < 4; ++c) rr[u * CPL + c] = ln_affine(rr[u * CPL + c], mean, rstd, lg[c], lb[c]);
                        }
                    }
                }
                if constexpr (FOLD) {
                    // acc <- rstd_row * (acc - mean_row * c_col) + bias_col, the LayerNorm of the A row applied to the product, as
                    // two packed FMAs per pair of values: x = t * c + bias (t = -mean * rstd), acc * rstd + x.  The row's
                    // (mean, rstd) sit in the padding of its staged row (stage): one 8-byte LDS read per instruction.
#pragma unroll
                    for (int u = 0; u < GROUP; ++u) {
                        const float2 st = *(const float2*)(region + ((t0 + u) * RPI + rsub) * G4D_EPI_STRIDE + 128);
                        const float t = -st.x * st.y;
                        const f32x2 t2 = {t, t}, r2 = {st.y, st.y};
#pragma unroll
                        for (int c = 0; c < CPL; c += 2) {
                            const f32x2 x = __builtin_elementwise_fma(t2, f32x2{fc[c], fc[c + 1]}, f32x2{bias[c], bias[c + 1]});
                            const f32x2 y = __builtin_elementwise_fma(f32x2{v[u * CPL + c], v[u * CPL + c + 1]}, r2, x);
                            v[u * CPL + c] = y.x; v[u * CPL + c + 1] = y.y;
                            bb[u * CPL + c] = 0.f; bb[u * CPL + c + 1] = 0.f;
                        }
                    }
                }
                epi_values<ACT, RES, SCALE, NV>(v, bb, rr, ss, hh);
                if constexpr (CHK16) {
#pragma unroll
                    for (int x = 0; x < NV; ++x) bad |= out_of_range(v[x], LoRange<T>::limit);
                } else if (chk_final) {
#pragma unroll
                    for (int x = 0; x < NV; ++x) bad |= out_of_range(v[x], ZETT_F32_MAX);
                }
                if constexpr (LN16) {
                    // per-row (sum, sum of squares) over the wave's 128 columns, eight per lane: a reduce-scatter over the 16 lanes
                    // of a row — after the xor-8 and xor-4 exchanges a lane carries ONE of the group's four instructions' rows,
                    // after two butterfly steps its totals; lane (idx & 3) == group keeps them
                    static_assert(!LN16 || (GROUP == 4 && CPL == 8 && RPI == 4), "reduce-scatter layout");
                    float S[4], Q[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float* x = v + 8 * u;
                        S[u] = ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
                        Q[u] = __builtin_fmaf(x[0], x[0], __builtin_fmaf(x[1], x[1], __builtin_fmaf(x[2], x[2], __builtin_fmaf(x[3], x[3],
                               __builtin_fmaf(x[4], x[4], __builtin_fmaf(x[5], x[5], __builtin_fmaf(x[6], x[6], x[7] * x[7])))))));
                    }
                    const bool b3 = (idx & 8) != 0, b2 = (idx & 4) != 0;
                    float ks0 = b3 ? S[2] : S[0], ks1 = b3 ? S[3] : S[1], kq0 = b3 ? Q[2] : Q[0], kq1 = b3 ? Q[3] : Q[1];
                    ks0 += g4d_xor_lane<8>(b3 ? S[0] : S[2]); ks1 += g4d_xor_lane<8>(b3 ? S[1] : S[3]);
                    kq0 += g4d_xor_lane<8>(b3 ? Q[0] : Q[2]); kq1 += g4d_xor_lane<8>(b3 ? Q[1] : Q[3]);
                    float ks = b2 ? ks1 : ks0, kq = b2 ? kq1 : kq0;
                    ks += g4d_xor_lane<4>(b2 ? ks0 : ks1); kq += g4d_xor_lane<4>(b2 ? kq0 : kq1);
                    ks += g4d_xor_lane<2>(ks); kq += g4d_xor_lane<2>(kq);
                    ks += g4d_xor_lane<1>(ks); kq += g4d_xor_lane<1>(kq);
                    const bool mine = (idx & 3) == (t0 >> 2);
                    acc_s = mine ? ks : acc_s; acc_q = mine ? kq : acc_q;
                } else if constexpr (LNP) {
                    // per-row (sum, sum of squares) over the wave's 128 columns: a reduce-scatter over the 32 lanes of a row
                    // group — after the xor-16 and xor-8 exchanges a lane carries ONE of the group's four row pairs, after
                    // three butterfly steps the row's totals; lane (idx & 7) == group keeps them
                    static_assert(GROUP == 4 && CPL == 4 && RPI == 2, "reduce-scatter layout");
                    float S[4], Q[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float a = v[4 * u], b = v[4 * u + 1], c = v[4 * u + 2], d = v[4 * u + 3];
                        S[u] = (a + b) + (c + d);
                        Q[u] = __builtin_fmaf(a, a, __builtin_fmaf(b, b, __builtin_fmaf(c, c, d * d)));
                    }
                    const bool b4 = (idx & 16) != 0, b3 = (idx & 8) != 0;
                    float ks0 = b4 ? S[2] : S[0], ks1 = b4 ? S[3] : S[1], kq0 = b4 ? Q[2] : Q[0], kq1 = b4 ? Q[3] : Q[1];
                    ks0 += g4d_xor_lane<16>(b4 ? S[0] : S[2]); ks1 += g4d_xor_lane<16>(b4 ? S[1] : S[3]);
                    kq0 += g4d_xor_lane<16>(b4 ? Q[0] : Q[2]); kq1 += g4d_xor_lane<16>(b4 ? Q[1] : Q[3]);
                    float ks = b3 ? ks1 : ks0, kq = b3 ? kq1 : kq0;
                    ks += g4d_xor_lane<8>(b3 ? ks0 : ks1); kq += g4d_xor_lane<8>(b3 ? kq0 : kq1);
                    ks += g4d_xor_lane<4>(ks); kq += g4d_xor_lane<4>(kq);
                    ks += g4d_xor_lane<2>(ks); kq += g4d_xor_lane<2>(kq);
                    ks += g4d_xor_lane<1>(ks); kq += g4d_xor_lane<1>(kq);
                    const bool mine = (idx & 7) == (t0 >> 2);
                    acc_s = mine ? ks : acc_s; acc_q = mine ? kq : acc_q;
                }
#pragma unroll
                for (int u = 0; u < GROUP; ++u) {
                    const int grow = grow0 + p * 64 + (t0 + u) * RPI;
                    if (!FULL && (grow >= g.M || !col_ok)) continue;
                    if constexpr (LO || LN16) {
                        const float4 a = make_float4(v[u * 8], v[u * 8 + 1], v[u * 8 + 2], v[u * 8 + 3]);
                        const float4 b = make_float4(v[u * 8 + 4], v[u * 8 + 5], v[u * 8 + 6], v[u * 8 + 7]);
                        store_out8<T>(e.out_lo + (size_t)grow * e.ld_lo + gcol, a, b);
                    } else {
                        const f32x4 x = {v[u * 4], v[u * 4 + 1], v[u * 4 + 2], v[u * 4 + 3]};
                        if constexpr (WF32) {
                            float* d = e.out_f32 + (size_t)grow * e.ld_f32 + gcol;
                            if (RES) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" :: "v"(d), "v"(x) : "memory");
                            else *(f32x4*)d = x;
                        }
                        if constexpr (EPI == G4D_EPI_BOTH || LNP)      // the same four values as the next GEMM's operand: 8 bytes per lane, 256 per row
                            store_out4<T>(e.out_lo + (size_t)grow * e.ld_lo + gcol, make_float4(x[0], x[1], x[2], x[3]));
                    }
                }
                // LN16: the first half of pass 0's residual registers is free once its first two groups are through — pass 1's rows move in
                if constexpr (LN16 && GEO::NPASS == 2) { if (p == 0 && t0 == GROUP) load_res16(1, 0, NIT / 2); }
            }
            if constexpr (LN16 && GEO::NPASS == 2) { if (p == 0) load_res16(1, NIT / 2, NIT); }      // (the second half of pass 1's residual rows)
            if constexpr (LNP) {
                // the lane holds row ((4 * (idx & 7) + 2 * b4 + b3) * 2 + rsub) of the pass: 64 lanes, 64 rows, one 512-byte store
                // (LN16: row ((4 * (idx & 3) + 2 * b3 + b2) * 4 + rsub))
                const int rp = LN16 ? (((4 * (idx & 3) + 2 * ((idx >> 3) & 1) + ((idx >> 2) & 1)) << 2) + rsub)
                                    : (((4 * (idx & 7) + 2 * ((idx >> 4) & 1) + ((idx >> 3) & 1)) << 1) + rsub);
                const int grow = m0 + wm * GEO::WROWS + p * 64 + rp;
                if (grow < g.M && n0 + wn * 128 < g.N)
                    e.stats_part[(size_t)((n0 + wn * 128) >> 7) * e.ld_part + grow] = make_float2(acc_s, acc_q);
            }
        };
        auto drain_pass = [&](int p) {
            const bool full = m0 + wm * GEO::WROWS + p * 64 + 64 <= g.M && n0 + wn * 128 + 128 <= g.N;
            if (full) drain(p, std::integral_constant<bool, true>{});
            else drain(p, std::integral_constant<bool, false>{});
        };
        if constexpr (LN16) { if (!res16_early) load_res(0); }
        else load_res(0);
        stage(0);
        __builtin_amdgcn_sched_barrier(0);         // (pass 1's residual registers only exist once pass 0's accumulators are staged)
        if constexpr (!LNP && GEO::NPASS == 2) load_res(1);
        __builtin_amdgcn_sched_barrier(0);
        drain_pass(0);
        if constexpr (GEO::NPASS == 2) {          // (the HALF tile is one pass of 64 rows per wave)
            if constexpr (LNP && !LN16) { __builtin_amdgcn_sched_barrier(0); load_res(1); __builtin_amdgcn_sched_barrier(0); }
            stage(1);
            drain_pass(1);
        }
        range_report(e.range_flag, bad, W16 ? ZETT_RANGE_BIT_ACTIVATION : ZETT_RANGE_BIT_OUTPUT);
    }
}

// Which epilogue a launch gets.  force_generic: zett_set_option("gemm_variant", 8) (A/B and the bit-identity test).
template <typename T>
inline int gemm4d_epi_mode(const GemmArgs<T>& g) {
    const GemmEpilogue<T>& e = g.epi;
    if (e.stats_part && e.residual_lo) {    // LayerNorm producer on the 16-bit residual stream
        if (sizeof(T) == 2 && e.out_lo && !e.out_f32 && !e.residual && e.act == ACT_NONE && !e.scale && !e.shift && !e.out_f32_b && e.split_col >= g.N &&
            g.N % 128 == 0 && e.ld_lo % 8 == 0 && e.ld_res_lo % 8 == 0) return G4D_EPI_LN16;
        return -1;
    }
    if (e.stats_part) {    // LayerNorm producer: only these instantiations write the partial statistics
        const bool common = e.out_lo && e.residual && !e.scale && !e.shift && !e.out_f32_b && e.split_col >= g.N && g.N % 128 == 0 &&
                            e.ld_lo % 4 == 0 && e.ld_res % 4 == 0;
        if (common && e.out_f32 && e.act == ACT_NONE && e.ld_f32 % 4 == 0) return G4D_EPI_F32_LN;
        if (common && !e.out_f32 && e.act == ACT_GELU_TANH && !e.res_stats && !e.res_index) return G4D_EPI_LO_LN;
        return -1;
    }
    if (e.fold_stats) {    // LayerNorm consumer
        if (e.fold_c && e.out_lo && !e.out_f32 && !e.residual && !e.scale && !e.shift && !e.out_f32_b && e.split_col >= g.N &&
            g.N % 8 == 0 && e.ld_lo % 8 == 0) return G4D_EPI_LO_FOLD;
        if (e.fold_c && e.out_f32 && !e.out_lo && !e.residual && e.scale && e.shift && e.act == ACT_NONE && !e.out_f32_b && e.split_col >= g.N &&
            g.N % 4 == 0 && e.ld_f32 % 4 == 0) return G4D_EPI_F32_SCALE_FOLD;
        return -1;
    }
    if (e.out_f32_b || e.split_col < g.N) return G4D_EPI_GENERIC;
    if (e.res_stats && (e.act != ACT_NONE || !e.residual)) return G4D_EPI_GENERIC;
    if (e.out_lo && !e.out_f32 && !e.residual && !e.scale && !e.shift && g.N % 8 == 0 && e.ld_lo % 8 == 0) return G4D_EPI_LO;
    if (e.out_f32 && e.out_lo && !e.residual && !e.scale && !e.shift && e.act == ACT_NONE && g.N % 4 == 0 && e.ld_f32 % 4 == 0 && e.ld_lo % 4 == 0)
        return G4D_EPI_BOTH;
    if (e.out_f32 && !e.out_lo && g.N % 4 == 0 && e.ld_f32 % 4 == 0 && (!e.residual || e.ld_res % 4 == 0)) {
        if (e.scale && e.shift && !e.residual && e.act == ACT_NONE) return G4D_EPI_F32_SCALE;
        // (instantiated: no activation with or without the residual, tanh-GELU with it; anything else is generic)
        if (!e.scale && !e.shift && (e.act == ACT_NONE || (e.act == ACT_GELU_TANH && e.residual))) return G4D_EPI_F32;
    }
    return G4D_EPI_GENERIC;
}

template <typename T, int ACT, bool RES, int EPI, bool HALF = false>
inline hipError_t launch_gemm4d_inst(const GemmArgs<T>& g, hipStream_t stream) {
    constexpr int lds = EPI == G4D_EPI_GENERIC ? G256_LDS_BYTES : (G4dGeom<HALF>::LDS_BYTES > G4D_LDS_BYTES ? G4dGeom<HALF>::LDS_BYTES : G4D_LDS_BYTES);
    static_assert(lds <= 160 * 1024, "LDS budget");
    static DeviceFlags attr;
    bool* done = attr.current();
    if (!done || !*done) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm4d_tn_kernel<T, ACT, RES, EPI, HALF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        if (done) *done = true;
    }
    constexpr int BM = HALF ? 128 : G256_BM;
    const int tiles_m = (g.M - g.row0 + BM - 1) / BM;
    const int tiles_n = (g.N + G256_BN - 1) / G256_BN;
    if (tiles_m <= 0 || tiles_n <= 0) return hipSuccess;
    hipLaunchKernelGGL((gemm4d_tn_kernel<T, ACT, RES, EPI, HALF>), dim3(tiles_m * tiles_n), dim3(256), lds, stream, g);
    return hipGetLastError();
}

// The instantiation a launch gets; HALF: the 128x256 tile on rows [g.row0, g.M) (streamlined epilogues only: a launch whose
// epilogue is the generic drain is never split).
template <typename T, int ACT, bool HALF>
inline hipError_t launch_gemm4d_act(const GemmArgs<T>& g, hipStream_t stream, int mode) {
    const bool res = g.epi.residual != nullptr;
    switch (mode) {
        case G4D_EPI_LO: return launch_gemm4d_inst<T, ACT, false, G4D_EPI_LO, HALF>(g, stream);
        case G4D_EPI_LO_FOLD: return launch_gemm4d_inst<T, ACT, false, G4D_EPI_LO_FOLD, HALF>(g, stream);
        case G4D_EPI_F32:
            if constexpr (ACT == ACT_NONE) return res ? launch_gemm4d_inst<T, ACT, true, G4D_EPI_F32, HALF>(g, stream) : launch_gemm4d_inst<T, ACT, false, G4D_EPI_F32, HALF>(g, stream);
            else if constexpr (ACT == ACT_GELU_TANH) { if (res) return launch_gemm4d_inst<T, ACT, true, G4D_EPI_F32, HALF>(g, stream); }
            [[fallthrough]];
        default:
            if constexpr (HALF) return hipErrorInvalidValue;
            else return res ? launch_gemm4d_inst<T, ACT, true, G4D_EPI_GENERIC>(g, stream) : launch_gemm4d_inst<T, ACT, false, G4D_EPI_GENERIC>(g, stream);
    }
}

template <typename T, bool HALF>
inline hipError_t launch_gemm4d_mode(const GemmArgs<T>& g, hipStream_t stream, int mode) {
    if (mode == G4D_EPI_F32_LN) return launch_gemm4d_inst<T, ACT_NONE, true, G4D_EPI_F32_LN, HALF>(g, stream);
    if (mode == G4D_EPI_LN16) return launch_gemm4d_inst<T, ACT_NONE, true, G4D_EPI_LN16, HALF>(g, stream);
    if (mode == G4D_EPI_LO_LN) return launch_gemm4d_inst<T, ACT_GELU_TANH, true, G4D_EPI_LO_LN, HALF>(g, stream);
    if (mode == G4D_EPI_F32_SCALE_FOLD) return launch_gemm4d_inst<T, ACT_NONE, false, G4D_EPI_F32_SCALE_FOLD, HALF>(g, stream);
    if (mode == G4D_EPI_F32_SCALE) return launch_gemm4d_inst<T, ACT_NONE, false, G4D_EPI_F32_SCALE, HALF>(g, stream);
    if (mode == G4D_EPI_BOTH) return launch_gemm4d_inst<T, ACT_NONE, false, G4D_EPI_BOTH, HALF>(g, stream);
    switch (g.epi.act) {
        case ACT_GELU_TANH: return launch_gemm4d_act<T, ACT_GELU_TANH, HALF>(g, stream, mode);
        case ACT_GELU_ERF: return launch_gemm4d_act<T, ACT_GELU_ERF, HALF>(g, stream, mode);
        default: return launch_gemm4d_act<T, ACT_NONE, HALF>(g, stream, mode);
    }
}

// true when `mode` (gemm4d_epi_mode) has a HALF instantiation for this launch
template <typename T>
inline bool gemm4d_half_ok(const GemmArgs<T>& g, int mode) {
    if (mode <= G4D_EPI_GENERIC) return false;
    if (mode == G4D_EPI_F32) return g.epi.act == ACT_NONE || (g.epi.act == ACT_GELU_TANH && g.epi.residual);
    return true;
}

// Rows [0, row0) of a launch go to 256x256 tiles, rows [row0, M) to 128x256 tiles (see the kernel's header): row0 = M (no
// split) unless the full tiles leave a partly filled last round of `cus` workgroups that the half tiles fill more cheaply.  A
// half tile is priced at 0.64 of a full one (half the MFMAs at ~0.78 of the rate).
inline int gemm4d_row_split(int M, int N, int cus = 256, double half_cost = 0.62) {
    const long tn = (N + G256_BN - 1) / G256_BN, tm = (M + G256_BM - 1) / G256_BM;
    const long T = tm * tn;
    if (T % cus == 0) return M;
    const long rounds = (T + cus - 1) / cus;
    // full tiles for as many row tiles as fit into R = rounds - 1 whole rounds
    const long tm_full = std::min<long>(tm, ((rounds - 1) * cus) / tn);
    const long row0 = tm_full * G256_BM;
    if (row0 >= M) return M;
    const long th = ((M - row0 + 127) / 128) * tn;
    const double cost = (double)((tm_full * tn + cus - 1) / cus) + half_cost * (double)((th + cus - 1) / cus);
    return cost < (double)rounds - 0.05 ? (int)row0 : M;
}

hipError_t launch_gemm_4d_half(const GemmArgs<f16_t>& g, hipStream_t stream, int mode);
hipError_t launch_gemm_4d_half(const GemmArgs<bf16_t>& g, hipStream_t stream, int mode);

template <typename T>
inline hipError_t launch_gemm4d(const GemmArgs<T>& g, hipStream_t stream, bool force_generic = false) {
    const int mode = (force_generic && !g.epi.stats_part && !g.epi.fold_stats) ? G4D_EPI_GENERIC : gemm4d_epi_mode(g);
    if (mode < 0) return hipErrorInvalidValue;       // a LayerNorm-fold launch whose outputs no instantiation carries
    // g.row0 on entry: -1 = let the launcher cut the launch (gemm4d_row_split), 0 = full tiles only, > 0 = the caller's cut (a multiple
    // of 256), -2 = half tiles only (tests and A/Bs)
    int split = g.row0 == -1 ? gemm4d_row_split(g.M, g.N) : (g.row0 == -2 ? 0 : (g.row0 > 0 ? std::min(g.row0, g.M) : g.M));
    if (split < g.M && (!gemm4d_half_ok(g, mode) || split % G256_BM != 0)) split = g.M;
    GemmArgs<T> full = g;
    full.row0 = 0;
    if (split < g.M) {
        full.M = split;
        if (split > 0)
            if (hipError_t e = launch_gemm4d_mode<T, false>(full, stream, mode); e != hipSuccess) return e;
        GemmArgs<T> rest = g;
        rest.row0 = split;
        return launch_gemm_4d_half(rest, stream, mode);          // (the HALF instantiations live in their own translation units: gemm4dh_<type>.hip)
    }
    return launch_gemm4d_mode<T, false>(full, stream, mode);
}

}  // namespace zett
