// gemm4p.hip.h — EXPERIMENT (r2), not part of the product: gemm4d (256x256 tile, four waves, both operands direct-to-LDS, 16x16x32 MFMAs on AGPR
// accumulators) as a PERSISTENT kernel: one workgroup per CU walks its tiles b, b + G, b + 2G, ... of the same
// XCD-aware order, and the operand pipeline never drains between them.
//
// Why.  The per-tile timeline of gemm4d (tools/experiments/trace4d.hip, K = 4096: 80-88 us of K loop) shows 2.0-2.2 us
// of prologue (the first two K steps requested, nothing to multiply yet), 0.7-1.5 us between the end of one workgroup
// and the start of the next on the same CU, and 0.5-0.8 us at the head of the epilogue (bias loads, barrier) — 3-4 %
// of a tile that no epilogue tuning reaches.  Here the LDS-DMA requests of the NEXT tile's K steps 0 and 1 ride in
// the request slots of the last two K steps of the current tile (which a one-tile kernel leaves empty), land while
// the epilogue runs, and the next tile's first MFMA follows the last store of the epilogue after one barrier and
// sixteen fragment reads.
//
// What makes that possible:
//   * the epilogue stages the accumulators through the 32 KiB of LDS the two 64 KiB operand stages leave free
//     (8 KiB per wave = 16 rows x 128 fp32, eight passes per tile instead of two), so the operand images of the
//     next tile can fill while it runs.  Rows are XOR-swizzled by their parity at 16-byte granularity instead of
//     padded (no room): both read patterns of gemm4d's epilogue stay conflict-free;
//   * tile edges are handled by the buffer descriptors (num_records ends at the last valid row: rows past the edge
//     read as zero and are never stored), so the lanes' request offsets do not depend on the tile and the next
//     tile costs eight SGPRs, not sixteen VGPRs;
//   * the accumulators are never zeroed: the first 64 MFMAs of a tile take the constant 0 as their C operand.
//
// Outcome (tools/experiments/trace4d.hip with PERSISTENT=1, K = 4096, per tile): the prologue (2.0-2.4 us) and most of the
// gap (0.8-1.4 -> 0.4 us) do disappear, but the K loop of a tile grows by 2-4 us — the fragment reads at its head are
// no longer hidden, and because vmcnt counts loads and stores in ONE in-order counter the second K step's wait for its
// operands also waits for the epilogue's stores to be acknowledged (in the one-tile kernel those drain under the next
// workgroup's own counter).  Net: plain bf16 epilogue 90.0 vs 89.9 us per tile, erf-GELU 94.9 vs 94.6, fp32 residual
// 106.3 vs 108.5; launch rates 1447 vs 1484, 1361 vs 1387, 1210 vs 1203 TFLOP/s; the headline step 56.1 vs 55.7 ms.
// Bit-identical to gemm4d (tile-variant test) and spill-free, but not faster: the product stays on the one-tile kernel.
// Two compiler facts found on the way: lambdas called from many unrolled sites must be always_inline (otherwise the
// closure, and with it every captured array, lives in scratch), and a __amdgpu_buffer_rsrc_t may not be a lambda
// parameter or the operand of ?: — the HOST pass then drops the kernel's stub without a diagnostic.
//
// Everything else — K loop schedule, reduction order, epilogue arithmetic (epi_values), residual handling (all loads
// before the first store), LayerNorm'd residuals, non-temporal fp32 stores — is gemm4d's: identical bits (tile-variant
// test).  Only the streamlined epilogues exist here; anything else (generic drain, K < 256, tile_order 1) stays on gemm4d.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "gemm4d.hip.h"

namespace zett {

constexpr int G4P_EPI_OFFSET = G256_LDS_BYTES;              // the epilogue regions start behind the two operand stages
constexpr int G4P_EPI_REGION = 16 * 128 * 4;                // 8 KiB per wave: 16 rows x 128 fp32
constexpr int G4P_LDS_BYTES = G4P_EPI_OFFSET + 4 * G4P_EPI_REGION;
static_assert(G4P_LDS_BYTES == 160 * 1024, "all of LDS");

template <typename T> __device__ __forceinline__ void mfma16_agpr_zero(f32x4& c, const u32x4& a, const u32x4& b);
template <> __device__ __forceinline__ void mfma16_agpr_zero<bf16_t>(f32x4& c, const u32x4& a, const u32x4& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
}
template <> __device__ __forceinline__ void mfma16_agpr_zero<f16_t>(f32x4& c, const u32x4& a, const u32x4& b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
}

template <typename T, int ACT = ACT_NONE, bool RES = false, int EPI = G4D_EPI_LO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm4p_tn_kernel(GemmArgs<T> g) {
    static_assert(EPI != G4D_EPI_GENERIC, "streamlined epilogues only");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BK = GEMM_ROW_BYTES / (int)sizeof(T);

    const int tiles_m = (g.M + G256_BM - 1) / G256_BM;
    const int tiles_n = (g.N + G256_BN - 1) / G256_BN;
    const int nwg = tiles_m * tiles_n;
    const int nk = g.K / BK;                                       // >= 4 (launcher)

    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);     // 0..3
    const int wm = wave >> 1, wn = wave & 1;

    // virtual workgroup id -> tile, the order of gemm4d (tile_order 0): XCD x owns a contiguous range of the order, which
    // walks groups of 4 column tiles row tile by row tile
    auto tile_of = [&](int vb, int& m0, int& n0) __attribute__((always_inline)) {
        const int q = nwg >> 3, r = nwg & 7, xcd = vb & 7;
        const int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
        constexpr int GROUP_N = 4;
        const int group_size = GROUP_N * tiles_m;
        const int first_n = (w / group_size) * GROUP_N;
        const int gn = (tiles_n - first_n) < GROUP_N ? (tiles_n - first_n) : GROUP_N;
        n0 = (first_n + (w % group_size) % gn) * G256_BN;
        m0 = ((w % group_size) / gn) * G256_BM;
    };
    // operand panel of a tile: base at its first row, num_records up to its last valid row (reads past it return 0)
    auto a_panel = [&](int m0) __attribute__((always_inline)) {
        const int rows = g.M - m0 < G256_BM ? g.M - m0 : G256_BM;
        return __builtin_amdgcn_make_buffer_rsrc((void*)(g.A + (size_t)m0 * g.lda), (short)0, rows * g.lda * (int)sizeof(T), G4R_RSRC_WORD3);
    };
    auto w_panel = [&](int n0) __attribute__((always_inline)) {
        const int rows = g.N - n0 < G256_BN ? g.N - n0 : G256_BN;
        return __builtin_amdgcn_make_buffer_rsrc((void*)(g.W + (size_t)n0 * g.ldw), (short)0, rows * g.ldw * (int)sizeof(T), G4R_RSRC_WORD3);
    };

    // request r (0..7) of a wave moves rows wave*64 + r*8 + lane/8 of an operand; the lane's LDS slot is chunk lane%8 of
    // its row, which holds source chunk (lane%8) ^ swz(row).  swz(row) = (row >> 1) & 7 only depends on the parity of r,
    // so a lane keeps two offsets per operand (r = 0, 1) and adds (r >> 1) * 16 rows in front of each request: the K loop
    // sits at the 256-VGPR limit and this kernel carries a few values across it that the one-tile kernel does not.
    uint32_t a_voff2[2], w_voff2[2];
    const uint32_t a_step16 = 16u * (uint32_t)g.lda * (uint32_t)sizeof(T), w_step16 = 16u * (uint32_t)g.ldw * (uint32_t)sizeof(T);
    unsigned char* const my_rows = smem + wave * 64 * GEMM_ROW_BYTES;
    // (the base passes through an empty volatile asm statement: as plain C++ the sum is loop-invariant and gets hoisted
    //  back into sixteen registers)
    auto row_offset = [](uint32_t base, uint32_t add) __attribute__((always_inline)) {
        asm volatile("" : "+v"(base));
        return base + add;
    };
    // descriptors of the current and of the next tile's panels (a buffer descriptor cannot be a lambda PARAMETER: the
    // host pass rejects the type there and silently drops the kernel; the lambdas pick one by flag)
    int vb = blockIdx.x;
    int m0, n0;
    tile_of(vb, m0, n0);
    __amdgpu_buffer_rsrc_t a_cur = a_panel(m0), w_cur = w_panel(n0);
    __amdgpu_buffer_rsrc_t a_nxt = a_cur, w_nxt = w_cur;
    auto dma_a = [&](bool next, int stage, int kt, int r) __attribute__((always_inline)) {
        const lds_ptr_t dst = (lds_ptr_t)(my_rows + stage * G256_STAGE_BYTES + r * 8 * GEMM_ROW_BYTES);
        const uint32_t voff = row_offset(a_voff2[r & 1], (uint32_t)(r >> 1) * a_step16);
        if (next) __builtin_amdgcn_raw_ptr_buffer_load_lds(a_nxt, dst, 16, voff, kt * GEMM_ROW_BYTES, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(a_cur, dst, 16, voff, kt * GEMM_ROW_BYTES, 0, 0);
    };
    auto dma_w = [&](bool next, int stage, int kt, int r) __attribute__((always_inline)) {
        const lds_ptr_t dst = (lds_ptr_t)(my_rows + stage * G256_STAGE_BYTES + G256_OPERAND_BYTES + r * 8 * GEMM_ROW_BYTES);
        const uint32_t voff = row_offset(w_voff2[r & 1], (uint32_t)(r >> 1) * w_step16);
        if (next) __builtin_amdgcn_raw_ptr_buffer_load_lds(w_nxt, dst, 16, voff, kt * GEMM_ROW_BYTES, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(w_cur, dst, 16, voff, kt * GEMM_ROW_BYTES, 0, 0);
    };

    f32x4 acc[8][8];                 // 128x128 per wave as 8x8 tiles of 16x16 (first written by the C = 0 MFMAs of a tile)
    int a_off[2], w_off[2];
    // (the lanes' request and fragment offsets are recomputed at the top of every tile from an opaque copy of the thread
    //  id: six VGPRs that would otherwise have to survive the epilogue, which wants every register it can get)
    auto lane_constants = [&]() __attribute__((always_inline)) {
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));
        const int ln = t & 63;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int row = wave * 64 + r * 8 + (ln >> 3);
            const int chunk = ((ln & 7) ^ ((row >> 1) & 7)) << 4;
            a_voff2[r] = (uint32_t)row * (uint32_t)g.lda * (uint32_t)sizeof(T) + chunk;
            w_voff2[r] = (uint32_t)row * (uint32_t)g.ldw * (uint32_t)sizeof(T) + chunk;
        }
        const int l15 = ln & 15, kq = ln >> 4;
        const int swz = (l15 >> 1) & 7;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int c = ((kb * 4 + kq) ^ swz) << 4;
            a_off[kb] = (wm * 128 + l15) * GEMM_ROW_BYTES + c;
            w_off[kb] = G256_OPERAND_BYTES + (wn * 128 + l15) * GEMM_ROW_BYTES + c;
        }
    };
    lane_constants();
    u32x4 fa[2][8], fw[2][8];
    auto read_a = [&](int stage, int kb, int i) __attribute__((always_inline)) {
        fa[kb][i] = *(const u32x4*)(smem + stage * G256_STAGE_BYTES + a_off[kb] + i * 16 * GEMM_ROW_BYTES);
    };
    auto read_w = [&](int stage, int kb, int j) __attribute__((always_inline)) {
        fw[kb][j] = *(const u32x4*)(smem + stage * G256_STAGE_BYTES + w_off[kb] + j * 16 * GEMM_ROW_BYTES);
    };

    int par = 0;                     // stage of K step 0 of the current tile (step kt lives in stage (kt + par) & 1)

    // ---- prologue of the first tile: steps 0 and 1 requested, step 0 landed
#pragma unroll
    for (int r = 0; r < 8; ++r) dma_w(false, 0, 0, r);
#pragma unroll
    for (int r = 0; r < 8; ++r) dma_a(false, 0, 0, r);
#pragma unroll
    for (int r = 0; r < 8; ++r) dma_w(false, 1, 1, r);
#pragma unroll
    for (int r = 0; r < 8; ++r) dma_a(false, 1, 1, r);
    __builtin_amdgcn_s_waitcnt(g4d_wait_vm(16));
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // K step of a wave (schedule of gemm4d.hip.h).  FIRST: the tile's first step (block 0 multiplies onto C = 0).
    // DMA: 1 = request step kt+2 of this tile, 2 = request step kt+2-nk of the NEXT tile (if there is one), 0 = nothing.
    // MORE: step kt+1 belongs to this tile (its block-0 fragments are read at the end of this step).
    bool has_next = false, later_tile = false;
    auto step = [&](int kt, auto first_c, auto dma_c, auto more_c, auto wave_c) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_c)::value, MORE = decltype(more_c)::value;
        constexpr int DMA = decltype(dma_c)::value, WV = decltype(wave_c)::value;
        const int cur = (kt + par) & 1;
        const bool dma_on = DMA == 1 || (DMA == 2 && has_next);
        const int dkt = DMA == 2 ? kt + 2 - nk : kt + 2;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int p = kb * 64 + i * 8 + j;
            if (p == 36 && DMA != 0) {      // this wave has every fragment of stage cur in registers: the stage may be refilled
                if (dma_on) {
                    __builtin_amdgcn_s_waitcnt(G4R_WAIT_LGKM0);
                    __builtin_amdgcn_s_barrier();
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // the 16 requests of this step may be in flight, those of the previous one not.  Not in the first step of a
            // tile that follows another: its step 1 landed before the epilogue's first store (every wave waited, then
            // the barrier), and the counter is in order — waiting here would wait for the epilogue's stores to be
            // acknowledged, which in a one-tile kernel is the NEXT workgroup's free ride
            if (p == 102 && MORE && !(FIRST && later_tile)) {
                if (dma_on) __builtin_amdgcn_s_waitcnt(g4d_wait_vm(16));
                else __builtin_amdgcn_s_waitcnt(g4d_wait_vm(0));
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            if (FIRST && kb == 0) mfma16_agpr_zero<T>(acc[i][j], fa[kb][i], fw[kb][j]);
            else mfma16_agpr<T>(acc[i][j], fa[kb][i], fw[kb][j]);
            if (p < 16 && (p & 1) == 0) read_w(cur, 1, p >> 1);
            if (p >= 16 && p <= 30 && (p & 1) == 0) read_a(cur, 1, (p - 16) >> 1);
            if (DMA != 0 && p >= 38 && p < 70 && WV == ((p - 38) & 3)) { if (dma_on) dma_w(DMA == 2, cur, dkt, (p - 38) >> 2); }
            if (DMA != 0 && p >= 70 && p < 102 && WV == ((p - 70) & 3)) { if (dma_on) dma_a(DMA == 2, cur, dkt, (p - 70) >> 2); }
            if (MORE && p >= 103 && p <= 110) read_w(cur ^ 1, 0, p - 103);
            if (MORE && p >= 111 && p <= 125 && (p & 1) == 1) read_a(cur ^ 1, 0, (p - 111) >> 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    typedef std::integral_constant<bool, true> yes_t;
    typedef std::integral_constant<bool, false> no_t;
    typedef std::integral_constant<int, 0> dma_none_t;
    typedef std::integral_constant<int, 1> dma_cur_t;
    typedef std::integral_constant<int, 2> dma_next_t;
    auto k_loop = [&](auto wave_c) __attribute__((always_inline)) {
        step(0, yes_t{}, dma_cur_t{}, yes_t{}, wave_c);
        int kt = 1;
        for (; kt + 2 < nk; ++kt) step(kt, no_t{}, dma_cur_t{}, yes_t{}, wave_c);
        step(kt, no_t{}, dma_next_t{}, yes_t{}, wave_c);
        step(kt + 1, no_t{}, dma_next_t{}, no_t{}, wave_c);
    };

    // ---- epilogue pieces (gemm4d.hip.h's streamlined epilogues on 16-row passes)
    constexpr bool LO = EPI == G4D_EPI_LO, SCALE = EPI == G4D_EPI_F32_SCALE;
    constexpr int CPL = LO ? 8 : 4;            // columns per lane
    constexpr int LPR = 128 / CPL;             // lanes per row
    constexpr int RPI = 64 / LPR;              // rows per wave instruction: 4 / 2
    constexpr int NIT = 16 / RPI;              // instructions per 16-row pass: 4 / 8
    static_assert(!(LO && RES), "the 16-bit-only epilogue has no residual");
    float* const region = (float*)(smem + G4P_EPI_OFFSET + wave * G4P_EPI_REGION);
    const GemmEpilogue<T>& e = g.epi;

    for (;;) {
        // the tile after this one (its panels are requested in the last two K steps)
        const int vb_next = vb + (int)gridDim.x;
        has_next = vb_next < nwg;
        int m0n = 0, n0n = 0;
        if (has_next) {
            tile_of(vb_next, m0n, n0n);
            a_nxt = a_panel(m0n);
            w_nxt = w_panel(n0n);
        }
#ifdef G4D_TRACE
        const unsigned long long tr0 = wall_clock64();
#endif
        // block-0 fragments of step 0 (landed: prologue, or the previous tile's epilogue waited for them)
        lane_constants();
#pragma unroll
        for (int j = 0; j < 8; ++j) read_w(par, 0, j);
#pragma unroll
        for (int i = 0; i < 8; ++i) read_a(par, 0, i);

        if (wave == 0) k_loop(std::integral_constant<int, 0>{});
        else if (wave == 1) k_loop(std::integral_constant<int, 1>{});
        else if (wave == 2) k_loop(std::integral_constant<int, 2>{});
        else k_loop(std::integral_constant<int, 3>{});

        // ---- epilogue of tile (m0, n0)
        asm volatile("s_nop 15\n\ts_nop 15");     // last MFMA (8 passes) -> first accumulator read
#ifdef G4D_TRACE
        const unsigned long long tr2 = wall_clock64();
#endif
        int tid_e = threadIdx.x;
        asm volatile("" : "+v"(tid_e));
        const int lane_e = tid_e & 63;
        const int l15 = lane_e & 15, kq = lane_e >> 4;
        const int idx = lane_e % LPR, rsub = lane_e / LPR;
        const int gcol = n0 + wn * 128 + idx * CPL;
        const bool col_ok = gcol < g.N;
        const int grow0 = m0 + wm * 128 + rsub;    // the lane's row in instruction t of pass q: grow0 + q*16 + t*RPI
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
        // LayerNorm'd residual: lane l holds the statistics of rows l and 64 + l of the wave's quadrant
        const bool res_ln = RES && ACT == ACT_NONE && e.res_stats != nullptr;
        float lg[4] = {1.f, 1.f, 1.f, 1.f}, lb[4] = {0.f, 0.f, 0.f, 0.f};
        float2 pst[2] = {make_float2(0.f, 1.f), make_float2(0.f, 1.f)};
        if constexpr (RES && ACT == ACT_NONE) {
            if (res_ln) {
                if (col_ok) {
                    const float4 a = *(const float4*)(e.res_gamma + gcol), b = *(const float4*)(e.res_beta + gcol);
                    lg[0] = a.x; lg[1] = a.y; lg[2] = a.z; lg[3] = a.w; lb[0] = b.x; lb[1] = b.y; lb[2] = b.z; lb[3] = b.w;
                }
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    int srow = m0 + wm * 128 + hh * 64 + lane_e; srow = srow < g.M ? srow : g.M - 1;
                    pst[hh] = *(const float2*)(e.res_stats + 2 * (size_t)srow);
                }
            }
        }
        // residual rows: passes 0-3 are requested before the first store leaves (a load issued behind a store cannot be
        // waited for without waiting for the store); pass q + 4 is requested behind the stores of pass q — four passes
        // before it is needed, by which time those stores have long been acknowledged — because the registers for it
        // only exist once the accumulators of pass q have left theirs
        float4 res[8][RES ? NIT : 1];
        auto load_res = [&](int q) __attribute__((always_inline)) {
            if constexpr (RES) {
#pragma unroll
                for (int t = 0; t < NIT; ++t) {
                    const int grow = grow0 + q * 16 + t * RPI;
                    res[q][t] = (grow < g.M && col_ok) ? *(const float4*)(e.residual + (size_t)grow * e.ld_res + gcol) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        };
        // staged element (row, col) of a pass lives at row*128 + ((col/4) ^ (row&1))*4 + col%4
        auto stage = [&](int q) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    region[(kq * 4 + r) * 128 + (((j * 4 + (l15 >> 2)) ^ (r & 1)) << 2) + (l15 & 3)] = acc[q][j][r];
        };
        auto drain = [&](int q, auto full_c) __attribute__((always_inline)) {
            constexpr bool FULL = decltype(full_c)::value;
            // values per lane and group: 32 for the plain epilogues, 16 beside the residual registers or a GELU's
            // temporaries (this kernel keeps ~40 VGPRs alive across the epilogue for the next tile's K loop)
            constexpr int GROUP = (RES || ACT != ACT_NONE) ? (LO ? 2 : 4) : NIT;
            constexpr int NV = GROUP * CPL;
#pragma unroll
            for (int t0 = 0; t0 < NIT; t0 += GROUP) {
                float v[NV], bb[NV], rr[NV], ss[NV], hh[NV];
#pragma unroll
                for (int u = 0; u < GROUP; ++u) {
                    const int lrow = (t0 + u) * RPI + rsub;
#pragma unroll
                    for (int c4 = 0; c4 < CPL; c4 += 4) {
                        const float4 x = *(const float4*)(region + lrow * 128 + ((((idx * CPL + c4) >> 2) ^ (lrow & 1)) << 2));
                        v[u * CPL + c4] = x.x; v[u * CPL + c4 + 1] = x.y; v[u * CPL + c4 + 2] = x.z; v[u * CPL + c4 + 3] = x.w;
                    }
#pragma unroll
                    for (int c = 0; c < CPL; ++c) { bb[u * CPL + c] = bias[c]; ss[u * CPL + c] = sc[c]; hh[u * CPL + c] = sh[c]; rr[u * CPL + c] = 0.f; }
                    if constexpr (RES) {
                        const float4 x = res[q][t0 + u];
                        rr[u * CPL] = x.x; rr[u * CPL + 1] = x.y; rr[u * CPL + 2] = x.z; rr[u * CPL + 3] = x.w;
                        if (ACT == ACT_NONE && res_ln) {
                            const int r0 = (q & 3) * 16 + (t0 + u) * RPI;      // rows r0 (rsub = 0) and r0 + 1 (rsub = 1) of the 64-row half q >> 2
                            const float2 st = pst[q >> 2];
                            const float mean0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, st.x), r0));
                            const float mean1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, st.x), r0 + 1));
                            const float rstd0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, st.y), r0));
                            const float rstd1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, st.y), r0 + 1));
                            const float mean = rsub ? mean1 : mean0, rstd = rsub ? rstd1 : rstd0;
#pragma unroll
                            for (int c = 0; c < 4; ++c) rr[u * CPL + c] = ln_affine(rr[u * CPL + c], mean, rstd, lg[c], lb[c]);
                        }
                    }
                }
                epi_values<ACT, RES, SCALE, NV>(v, bb, rr, ss, hh);
#pragma unroll
                for (int u = 0; u < GROUP; ++u) {
                    const int grow = grow0 + q * 16 + (t0 + u) * RPI;
                    if (!FULL && (grow >= g.M || !col_ok)) continue;
                    if constexpr (LO) {
                        const float4 a = make_float4(v[u * 8], v[u * 8 + 1], v[u * 8 + 2], v[u * 8 + 3]);
                        const float4 b = make_float4(v[u * 8 + 4], v[u * 8 + 5], v[u * 8 + 6], v[u * 8 + 7]);
                        store_out8<T>(e.out_lo + (size_t)grow * e.ld_lo + gcol, a, b);
                    } else {
                        float* d = e.out_f32 + (size_t)grow * e.ld_f32 + gcol;
                        const f32x4 x = {v[u * 4], v[u * 4 + 1], v[u * 4 + 2], v[u * 4 + 3]};
                        if (RES) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" :: "v"(d), "v"(x) : "memory");
                        else *(f32x4*)d = x;
                    }
                }
            }
        };
        const bool full = m0 + wm * 128 + 128 <= g.M && n0 + wn * 128 + 128 <= g.N;      // the wave's whole quadrant lies inside the matrix
        auto drain_pass = [&](int q) __attribute__((always_inline)) {
            if (full) drain(q, std::integral_constant<bool, true>{});
            else drain(q, std::integral_constant<bool, false>{});
        };
#pragma unroll
        for (int q = 0; q < 4; ++q) load_res(q);
        stage(0);
        // everything requested so far has landed before the first store leaves: the next tile's first two K steps
        // (requested during the last two K steps), bias, statistics, the residual rows of passes 0-3
        __builtin_amdgcn_s_waitcnt(GEMM_WAIT_VMCNT0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (q > 0) stage(q);
            drain_pass(q);
            if (q + 4 < 8) {
                __builtin_amdgcn_sched_barrier(0);
                load_res(q + 4);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#ifdef G4D_TRACE
        if (wave == 0 && (threadIdx.x & 63) == 0 && vb < 32768) {
            unsigned long long* o = g4d_trace + (size_t)vb * 8;
            o[0] = tr0; o[1] = tr0; o[2] = tr2; o[3] = wall_clock64(); o[4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
            o[5] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
        }
#endif
        if (!has_next) break;
        // every wave has passed its vmcnt(0): the next tile's K steps 0 and 1 are in LDS for all of them
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        later_tile = true;
        vb = vb_next; m0 = m0n; n0 = n0n;
        a_cur = a_nxt; w_cur = w_nxt;
        par = (par + nk) & 1;
    }
}

template <typename T, int ACT, bool RES, int EPI>
inline hipError_t launch_gemm4p_inst(const GemmArgs<T>& g, hipStream_t stream) {
    static DeviceFlags attr;
    static int n_cu[64] = {};
    bool* done = attr.current();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!done || !*done) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm4p_tn_kernel<T, ACT, RES, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, G4P_LDS_BYTES);
        if (e != hipSuccess) return e;
        e = hipDeviceGetAttribute(&n_cu[dev], hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return e;
        if (done) *done = true;
    }
    const int tiles_m = (g.M + G256_BM - 1) / G256_BM;
    const int tiles_n = (g.N + G256_BN - 1) / G256_BN;
    const int nwg = tiles_m * tiles_n;
    if (nwg <= 0) return hipSuccess;
    int grid = n_cu[dev] > 0 ? n_cu[dev] & ~7 : 256;          // one workgroup per CU (all of LDS); a multiple of the XCD count
    if (grid < 8) grid = 8;
    if (grid > nwg) grid = nwg;
    hipLaunchKernelGGL((gemm4p_tn_kernel<T, ACT, RES, EPI>), dim3(grid), dim3(256), G4P_LDS_BYTES, stream, g);
    return hipGetLastError();
}

// true if the launch can take the persistent kernel (otherwise: gemm4d)
template <typename T>
inline bool gemm4p_eligible(const GemmArgs<T>& g) {
    constexpr int BK = GEMM_ROW_BYTES / (int)sizeof(T);
    if (g.tile_order != 0 || g.K / BK < 4) return false;
    if ((long)G256_BM * g.lda * (long)sizeof(T) > 0x7fffffffL || (long)G256_BN * g.ldw * (long)sizeof(T) > 0x7fffffffL) return false;
    return gemm4d_epi_mode(g) != G4D_EPI_GENERIC;
}

template <typename T>
inline hipError_t launch_gemm4p(const GemmArgs<T>& g, hipStream_t stream) {
    const int mode = gemm4d_epi_mode(g);
    const bool res = g.epi.residual != nullptr;
    if (mode == G4D_EPI_F32_SCALE) return launch_gemm4p_inst<T, ACT_NONE, false, G4D_EPI_F32_SCALE>(g, stream);
    if (mode == G4D_EPI_LO) {
        switch (g.epi.act) {
            case ACT_GELU_TANH: return launch_gemm4p_inst<T, ACT_GELU_TANH, false, G4D_EPI_LO>(g, stream);
            case ACT_GELU_ERF: return launch_gemm4p_inst<T, ACT_GELU_ERF, false, G4D_EPI_LO>(g, stream);
            default: return launch_gemm4p_inst<T, ACT_NONE, false, G4D_EPI_LO>(g, stream);
        }
    }
    if (mode == G4D_EPI_F32) {
        if (g.epi.act == ACT_GELU_TANH && res) return launch_gemm4p_inst<T, ACT_GELU_TANH, true, G4D_EPI_F32>(g, stream);
        if (g.epi.act == ACT_NONE) return res ? launch_gemm4p_inst<T, ACT_NONE, true, G4D_EPI_F32>(g, stream) : launch_gemm4p_inst<T, ACT_NONE, false, G4D_EPI_F32>(g, stream);
    }
    return hipErrorInvalidValue;      // not eligible: the caller checks gemm4p_eligible first
}

}  // namespace zett
