// gemm2w.hip.h — EXPERIMENT (round 3), not in the product: 128x256 tile, four waves (64x128 each), both operands
// HBM/L2 -> LDS by buffer_load_dwordx4 ... lds, on v_mfma_f32_16x16x32_{bf16,f16}, built so that TWO workgroups are
// resident per CU (128 accumulators + 64 fragment registers per wave, 72 KiB of LDS per workgroup).
//
//   C[M,N] = epilogue( A[M,K] · W[N,K]ᵀ )      (contract and epilogue arithmetic of gemm.hip.h; bit-identical results:
//                                               tools/gemm_bench g2w / g2wg, maxdiff 0 on every shape and epilogue)
//
// The idea ("next" 1).  gemm4d (256x256, one workgroup per CU, 512 registers per wave) leaves its CU without
// MFMA work during every prologue, epilogue and hand-over: nothing else is resident.  Behind the 17-33 us K loops of the
// narrow hypernets (K = 768 / 1536) and on the partly filled rounds of a vocabulary shard that is a third to a half of a
// tile's time.  Two half-size tiles per CU whose phases drift apart would run one tile's epilogue under the other's MFMAs.
//
// The result: it loses on EVERY launch shape of the path (tools/g2w_sweep.sh, bf16, randn operands, one box,
// gpurun_out/r3b; TFLOP/s gemm4d -> gemm2w):
//     169283 x 2304 x  768  plain 16-bit   955 ->  805      fp32 residual  623 -> 564      erf-GELU  797 -> 724
//     169283 x  768 x 1536                1128 ->  872                     850 -> 712               1012 -> 837
//     118979 x 6144 x 2048                1306 -> 1027                    1072 -> 853               1189 -> 954
//       9700 x 12288 x 4096 (a shard)     1385 -> 1073                    1187 -> 935               1286 -> 1013
//      77450 x 4096 x 4096                1439 -> 1072                    1275 -> 970               1358 -> 1026
// Why, by ablation of the K loop (-DG2W_ABL, 77450 x 4096 x 4096, same box as a 1 507 TFLOP/s gemm4d):
//     bare MFMAs (no requests, no barrier, no fragment reads)      1 905 TFLOP/s
//     + barrier + fragment reads                                   1 707
//     + the LDS-DMA requests (= the kernel)                        1 095      (without the barrier: 1 095; without the reads: 1 118)
// The whole loss is the LDS-DMA requests.  A 1 KiB request holds the issue port of its SIMD for ~52 cycles (DESIGN.md,
// round-1 finding 1) and here NOTHING overlaps it: per wave and K step 6 requests x 52 = 312 cycles beside 512 cycles of
// MFMA, 1 / 1.61 = 0.62 of the request-free rate — measured 0.64.  The cost is per byte, and a 128x256 tile moves 1.5x
// the operand bytes per FLOP of a 256x256 one (gemm4d: 16 requests per 128 MFMAs = 0.41, of which about half hides).
// All-zero operands (no power limit) say the same: gemm4d 1 892, gemm2w 1 243.  Hiding a 5-20 us epilogue cannot pay
// for a K loop that is 28 % slower, at any K.  Register staging instead of LDS-DMA (gemm8r's loader: ~25-30 issue
// cycles per KiB) would need 32 more registers than two waves per SIMD have and would still sit at ~0.72; gemm8r
// itself (eight waves, one workgroup) is the measured form of that trade and is 4-12 % behind gemm4d.  On this chip
// the operand path, not occupancy, sets the tile: the largest tile wins, and what a lone workgroup cannot hide has
// to be made smaller instead.
// Also learnt on the way: (1) ds_read_b128 is served in four groups of 16 NON-consecutive lanes (MI355X_MICROARCH.md LDS
// table); a swizzle that is conflict-free for 16 consecutive lanes ((row >> 2) & 3 on 64-byte rows) is a two-way conflict
// on every 16x16x32 fragment read — g2w_swz below is the conflict-free one (worth 3 % here: the reads were never the
// bound); (2) peeled K-loop tails (8 per wave copy) made the register allocator spill 200 registers around their joins;
// wave-uniform run-time predicates in ONE step body cost two scalar branches per step and no spill.
//
// K step = 32 elements (64-byte LDS rows): per wave 4 x 8 MFMAs, 12 fragment reads (4 A, 8 W) and 6 of the
// workgroup's 24 LDS-DMA requests (A 8, W 16; 1 KiB = 16 rows each).  Three LDS stages of 24 KiB (A 8 + W 16); during
// step t the MFMAs run on the fragments of step t (registers), stage (t+1)%3 is read, stage (t+2)%3 is landing, stage
// t%3 is free and takes the requests of step t+3: a request has two steps to land.  One barrier per step, at its top.
// LDS image: 64-byte rows, the four 16-byte chunks of a row XOR-swizzled by g2w_swz(row); LDS-DMA writes are
// lane-linear, so the swizzle is applied to each lane's SOURCE chunk (as in gemm4d).
// The K reduction order per accumulator is ascending K in one fp32 chain: that of every other tile variant.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "gemm_tile.hip.h"
#include "gemm4d.hip.h"      // mfma16_agpr, g4d_wait_vm, g4d_xor_lane, the EPI modes

namespace zett {

constexpr int G2W_BM = 128, G2W_BN = 256;
constexpr int G2W_ROWB = 64;                                   // K bytes per LDS row
constexpr int G2W_A_BYTES = G2W_BM * G2W_ROWB;                 // 8 KiB
constexpr int G2W_W_BYTES = G2W_BN * G2W_ROWB;                 // 16 KiB
constexpr int G2W_STAGE_BYTES = G2W_A_BYTES + G2W_W_BYTES;     // 24 KiB
constexpr int G2W_STAGES = 3;
constexpr int G2W_EPI_ROWS = 32;                               // rows a wave stages per epilogue pass (two passes)
constexpr int G2W_EPI_REGION = G2W_EPI_ROWS * G4D_EPI_STRIDE * 4;   // 16.5 KiB per wave
constexpr int G2W_LDS_BYTES = G2W_STAGES * G2W_STAGE_BYTES;    // 72 KiB: two workgroups per CU (160 KiB)
static_assert(4 * G2W_EPI_REGION <= G2W_LDS_BYTES, "the epilogue stages inside the operand images");

// Chunk swizzle of the 64-byte LDS rows.  A ds_read_b128 is served in four groups of 16 lanes that are NOT consecutive
// lanes (MI355X_MICROARCH.md, LDS table): {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, {32-35, 44-47, 52-59},
// {36-43, 48-51, 60-63}.  With the fragment layout of the 16x16x32 MFMA (lane l: row l & 15, chunk l >> 4) every group
// holds all 16 rows, rows 4-11 with the chunk of the other rows ^ 1.  Four rows share a 256-byte bank row (row & 3 equal),
// so they need four distinct chunk slots: chunk ^ (row & 8 ? 3 : 0) gives {c, c^1, c^1^3, c^3} for rows r, r+4, r+8, r+12
// in every group — conflict-free.  (A swizzle by (row >> 2) & 3, which is conflict-free for 16 CONSECUTIVE lanes, is a
// two-way conflict on every fragment read.)
__device__ __forceinline__ int g2w_swz(int row) { return (row & 8) ? 3 : 0; }

template <typename T, int ACT = ACT_NONE, bool RES = false, int EPI = G4D_EPI_GENERIC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm2w_tn_kernel(GemmArgs<T> g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    static_assert(sizeof(T) == 2, "16-bit operands");
    constexpr int BK = G2W_ROWB / (int)sizeof(T);              // 32

    const int tiles_m = (g.M + G2W_BM - 1) / G2W_BM;
    const int tiles_n = (g.N + G2W_BN - 1) / G2W_BN;
    const int nwg = tiles_m * tiles_n;
    int wg = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
    }
    // tile order of gemm4d: groups of GROUP_N column tiles walked column-first, so the 64 tiles an XCD runs at a time
    // (32 CUs x 2) are 16 row tiles x 4 column tiles: the W panels stay in its L2, A streams
    const int GROUP_N = g.group > 0 ? g.group : 4;
    int tm, tn;
    {
        const int group_size = GROUP_N * tiles_m;
        const int first_n = (wg / group_size) * GROUP_N;
        const int gn = (tiles_n - first_n) < GROUP_N ? (tiles_n - first_n) : GROUP_N;
        tn = first_n + (wg % group_size) % gn;
        tm = (wg % group_size) / gn;
    }
    const int m0 = tm * G2W_BM, n0 = tn * G2W_BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // 0..3
    const int wm = wave >> 1, wn = wave & 1;

    // LDS-DMA requests: a request moves 16 rows (lane/4) x 64 B; lane's LDS slot is chunk lane%4 of its row, which holds
    // source chunk (lane%4) ^ swz(row).  Wave w issues A requests 2w, 2w+1 and W requests 4w .. 4w+3.
    const unsigned char* a_base = (const unsigned char*)(g.A + (size_t)m0 * g.lda);
    const unsigned char* w_base = (const unsigned char*)(g.W + (size_t)n0 * g.ldw);
    uint32_t a_voff[2], w_voff[4];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int row = (wave * 2 + r) * 16 + (lane >> 2);
        const int chunk = ((lane & 3) ^ g2w_swz(row)) << 4;
        int ar = row; ar = m0 + ar < g.M ? ar : g.M - 1 - m0;
        a_voff[r] = (uint32_t)ar * (uint32_t)g.lda * (uint32_t)sizeof(T) + chunk;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = (wave * 4 + r) * 16 + (lane >> 2);
        const int chunk = ((lane & 3) ^ g2w_swz(row)) << 4;
        int wr = row; wr = n0 + wr < g.N ? wr : g.N - 1 - n0;
        w_voff[r] = (uint32_t)wr * (uint32_t)g.ldw * (uint32_t)sizeof(T) + chunk;
    }
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, (short)0, 0x7fffffff, G4R_RSRC_WORD3);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)w_base, (short)0, 0x7fffffff, G4R_RSRC_WORD3);
    // request q of a step (0, 1: A; 2..5: W) into LDS stage `stage`
    auto dma = [&](int kt, int stage, int q) {
        unsigned char* S = smem + stage * G2W_STAGE_BYTES;
        if (q < 2)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_ptr_t)(S + (wave * 2 + q) * 16 * G2W_ROWB), 16, a_voff[q], kt * G2W_ROWB, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(S + G2W_A_BYTES + (wave * 4 + (q - 2)) * 16 * G2W_ROWB), 16, w_voff[q - 2], kt * G2W_ROWB, 0, 0);
    };

    f32x4 acc[4][8];                 // 64x128 per wave as 4x8 tiles of 16x16
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    // fragment of a 16x16x32 MFMA: lane l holds row (l & 15), K elements (l >> 4)*8 .. +7 = 16-byte chunk (l >> 4) of
    // the 64-byte row; 16-row steps leave the swizzle unchanged
    int a_off, w_off;
    {
        const int l15 = lane & 15, kq = lane >> 4;
        const int c = (kq ^ g2w_swz(l15)) << 4;
        a_off = (wm * 64 + l15) * G2W_ROWB + c;
        w_off = G2W_A_BYTES + (wn * 128 + l15) * G2W_ROWB + c;
    }
    // Registers: 128 accumulators (AGPRs) + the fragments must fit 256 per wave.  MFMA order of a step is column tile
    // (j) outer, row tile (i) inner: a W fragment dies after its four MFMAs and its registers take the same fragment of
    // the NEXT step right away (single-buffered W: 32 registers), the four A fragments live for the whole step and are
    // double-buffered (32 registers).
    u32x4 fa[2][4], fw[8];
    auto read_a = [&](int stage, int buf, int i) { fa[buf][i] = *(const u32x4*)(smem + stage * G2W_STAGE_BYTES + a_off + i * 16 * G2W_ROWB); };
    auto read_w = [&](int stage, int j) { fw[j] = *(const u32x4*)(smem + stage * G2W_STAGE_BYTES + w_off + j * 16 * G2W_ROWB); };

    const int nk = g.K / BK;
    // ---- prologue: steps 0, 1, 2 requested; step 0 landed; its fragments read
#pragma unroll
    for (int q = 0; q < 6; ++q) dma(0, 0, q);
    if (nk > 1) {
#pragma unroll
        for (int q = 0; q < 6; ++q) dma(1, 1, q);
    }
    if (nk > 2) {
#pragma unroll
        for (int q = 0; q < 6; ++q) dma(2, 2, q);
    }
    if (nk > 2) __builtin_amdgcn_s_waitcnt(g4d_wait_vm(12));
    else if (nk > 1) __builtin_amdgcn_s_waitcnt(g4d_wait_vm(6));
    else __builtin_amdgcn_s_waitcnt(g4d_wait_vm(0));
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) read_a(0, 0, i);
#pragma unroll
    for (int j = 0; j < 8; ++j) read_w(0, j);

    // One K step.  BUF: the A fragment buffer its MFMAs run on (the reads of step kt+1 fill the other one); WV: the wave
    // this copy of the loop belongs to — wave w issues its request q under MFMA 8 + 4q + w, so that the four waves never
    // queue at the texture path together.  One MFMA per scheduling region.
    //   p = 4j + i:  MFMA (i, j);  after p = 4j + 3 the W fragment j of step kt+1 is read into the registers that just
    //   became free; the A fragments of step kt+1 are read under p = 0 .. 3.
    // The ends of the K range are handled by wave-uniform run-time predicates, not by peeled copies of the step (eight
    // peeled tails per wave copy cost the register allocator its balance: it spilled around their joins): requests of
    // steps past the end are skipped by a scalar branch, the wait at the top counts what is really in flight, and the
    // fragment reads of a step that does not exist fetch stale LDS bytes that nothing consumes.
    auto step = [&](int kt, int s0, int s1, auto buf_c, auto wave_c) {
        constexpr int BUF = decltype(buf_c)::value;
        constexpr int WV = decltype(wave_c)::value;
        // top of the step: this wave's fragments of step kt are in registers, its requests of step kt+1 have landed
        // (those of step kt+2, if it exists, may still be in flight)
#ifndef G2W_ABL
#define G2W_ABL 0        // experiments (tools/gemm_bench -DG2W_ABL=bits): 1 no requests, 2 no barrier, 4 no fragment reads in the K loop
#endif
        if (kt + 2 < nk) __builtin_amdgcn_s_waitcnt(g4d_wait_vm(6) & G4R_WAIT_LGKM0);
        else __builtin_amdgcn_s_waitcnt(g4d_wait_vm(0) & G4R_WAIT_LGKM0);
        if (!(G2W_ABL & 2)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const bool more3 = kt + 3 < nk && !(G2W_ABL & 1);
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = j * 4 + i;
            mfma16_agpr<T>(acc[i][j], fa[BUF][i], fw[j]);
            if (p < 4 && !(G2W_ABL & 4)) read_a(s1, BUF ^ 1, p);
            if (i == 3 && !(G2W_ABL & 4)) read_w(s1, j);
            if (p >= 8 && ((p - 8) & 3) == WV) { if (more3) dma(kt + 3, s0, (p - 8) >> 2); }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    typedef std::integral_constant<int, 0> buf0_t;
    typedef std::integral_constant<int, 1> buf1_t;
    auto k_loop = [&](auto wave_c) {
        int s0 = 0, s1 = 1, s2 = 2;      // stages of steps kt, kt+1, kt+2
        for (int kt = 0; kt < nk; kt += 2) {          // nk is even (K % 64 == 0); the A fragment buffers alternate
            step(kt, s0, s1, buf0_t{}, wave_c);
            step(kt + 1, s1, s2, buf1_t{}, wave_c);
            const int t = s0; s0 = s2; s2 = s1; s1 = t;
        }
    };
    if (wave == 0) k_loop(std::integral_constant<int, 0>{});
    else if (wave == 1) k_loop(std::integral_constant<int, 1>{});
    else if (wave == 2) k_loop(std::integral_constant<int, 2>{});
    else k_loop(std::integral_constant<int, 3>{});

    asm volatile("s_nop 15\n\ts_nop 15");     // last MFMA (4 passes) -> first accumulator read
    // every wave has read its last fragments before the last step's barrier: the operand images are free, each wave
    // stages into its own region of them
    int tid_e = tid;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63;
    const int l15 = lane_e & 15, kq = lane_e >> 4;
    const GemmEpilogue<T>& e = g.epi;
    float* region = (float*)(smem + wave * G2W_EPI_REGION);
    auto stage_acc = [&](int sp) {          // rows sp*32 .. +31 of the wave's 64
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    region[(i2 * 16 + kq * 4 + r) * G4D_EPI_STRIDE + j * 16 + l15] = acc[2 * sp + i2][j][r];
    };

    if constexpr (EPI == G4D_EPI_GENERIC) {
        // ---- generic epilogue: EpiDrain of gemm_tile.hip.h on 16 x 128 staged values per pass (one 16-row MFMA tile row
        // at a time: the residual rows of a pass are 32 registers; row stride 128, which is what EpiDrain addresses)
        typedef EpiDrain<T, ACT, RES, 16, 128, !RES, false> Drain;
        const int gcol = n0 + wn * 128 + (lane_e % Drain::LPR) * 8;
        const bool col_ok = gcol < g.N;
        float4 bias8[2], sc8[2], sh8[2];
        Drain::load_cols(e, gcol, col_ok, bias8, sc8, sh8);
        float4 lng[2], lnb[2];
        Drain::load_ln_cols(e, gcol, col_ok, lng, lnb);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 oa[Drain::NIT], ob[Drain::NIT];
            const int row0 = m0 + wm * 64 + i * 16;
            Drain::load_res(g, row0, gcol, col_ok, lane_e, oa, ob);
            float2 lnst;
            Drain::load_res_stats(g, row0, lane_e, lnst);
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    region[(kq * 4 + r) * 128 + j * 16 + l15] = acc[i][j][r];
            __builtin_amdgcn_s_waitcnt(GEMM_WAIT_VMCNT0);
            Drain::ln_res(g, lane_e, lng, lnb, lnst, oa, ob);
            Drain::drain(g, region, row0, gcol, col_ok, lane_e, bias8, sc8, sh8, oa, ob);
        }
    } else {
        // ---- streamlined epilogues: gemm4d's arithmetic and lane mapping (so that the LayerNorm-fold statistics and
        // every output are the same bits), in passes of 32 rows with the residual rows of one pass in registers —
        // with a second workgroup resident the waits of this one are covered, and 128 VGPRs are all there is
        constexpr bool LO = EPI == G4D_EPI_LO || EPI == G4D_EPI_LO_FOLD, SCALE = EPI == G4D_EPI_F32_SCALE;
        constexpr bool LNP = EPI == G4D_EPI_F32_LN, FOLD = EPI == G4D_EPI_LO_FOLD;
        static_assert(!LNP || (RES && ACT == ACT_NONE), "the LayerNorm producer is the residual epilogue");
        static_assert(!(LO && RES), "the 16-bit-only epilogue has no residual");
        constexpr int CPL = LO ? 8 : 4;            // columns per lane
        constexpr int LPR = 128 / CPL;             // lanes per row
        constexpr int RPI = 64 / LPR;              // rows per wave instruction
        constexpr int NIT = G2W_EPI_ROWS / RPI;    // instructions per pass
        constexpr bool W16 = LO || EPI == G4D_EPI_BOTH || LNP;
        constexpr bool CHK16 = W16 && LoRange<T>::checked;
        const bool chk_final = !W16 && e.range_final != 0;
        bool bad = false;
        const int idx = lane_e % LPR, rsub = lane_e / LPR;
        const int gcol = n0 + wn * 128 + idx * CPL;
        const bool col_ok = gcol < g.N;
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
        const bool res_ln = RES && ACT == ACT_NONE && e.res_stats != nullptr;
        float lg[4] = {1.f, 1.f, 1.f, 1.f}, lb[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (RES && ACT == ACT_NONE) {
            if (res_ln && col_ok) {
                const float4 a = *(const float4*)(e.res_gamma + gcol), b = *(const float4*)(e.res_beta + gcol);
                lg[0] = a.x; lg[1] = a.y; lg[2] = a.z; lg[3] = a.w; lb[0] = b.x; lb[1] = b.y; lb[2] = b.z; lb[3] = b.w;
            }
        }
        const bool res_ix = RES && e.res_index != nullptr;
        float fc[FOLD ? CPL : 1];
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
        }
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
            const int prow0 = m0 + wm * 64 + sp * G2W_EPI_ROWS;       // first row of the pass
            const int grow0 = prow0 + rsub;                            // the lane's row in instruction t: grow0 + t*RPI
            // per-pass row data, lane l (< 32) holding row l of the pass: statistics of a LayerNorm'd residual, indices of
            // indexed residual rows, statistics of the fold consumer
            float2 pst = make_float2(0.f, 1.f), fst = make_float2(0.f, 1.f);
            int rix = 0;
            {
                int srow = prow0 + (lane_e & 31); srow = srow < g.M ? srow : g.M - 1;
                if constexpr (RES) {
                    if (res_ix) { rix = e.res_index[srow]; }
                    if (ACT == ACT_NONE && res_ln) pst = *(const float2*)(e.res_stats + 2 * (size_t)(res_ix ? rix : srow));
                }
                if constexpr (FOLD) fst = *(const float2*)(e.fold_stats + 2 * (size_t)srow);
            }
            // GROUP instructions at a time (16 values per lane: 128 VGPRs are all there is).  The residual rows of a group
            // are requested one group ahead, i.e. before the previous group's stores: loads and stores share one in-order
            // counter, so the wait for them does not wait for those stores
            constexpr int GROUP = LO ? (ACT == ACT_GELU_ERF ? 1 : 2) : 4;      // (erf-GELU: five temporaries per pair of values)
            constexpr int NV = GROUP * CPL;
            float4 res_next[RES ? GROUP : 1];
            auto load_res = [&](int t0) {
                if constexpr (RES) {
#pragma unroll
                    for (int u = 0; u < GROUP; ++u) {
                        const int t = t0 + u;
                        const int grow = grow0 + t * RPI;
                        size_t rrow = (size_t)grow;
                        if (res_ix) {
                            const int i0 = __builtin_amdgcn_readlane(rix, t * RPI), i1 = __builtin_amdgcn_readlane(rix, t * RPI + 1);
                            rrow = (size_t)(rsub ? i1 : i0);
                        }
                        res_next[u] = (grow < g.M && col_ok) ? *(const float4*)(e.residual + rrow * e.ld_res + gcol) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
            };
            load_res(0);
            stage_acc(sp);
            if constexpr (FOLD) { if (lane_e < G2W_EPI_ROWS) *(float2*)(region + lane_e * G4D_EPI_STRIDE + 128) = fst; }
            const bool full = prow0 + G2W_EPI_ROWS <= g.M && n0 + wn * 128 + 128 <= g.N;
            float acc_s = 0.f, acc_q = 0.f;
#pragma unroll
            for (int t0 = 0; t0 < NIT; t0 += GROUP) {
                float v[NV], bb[NV], rr[NV], ss[NV], hh[NV];
                float4 res[RES ? GROUP : 1];
                if constexpr (RES) {
#pragma unroll
                    for (int u = 0; u < GROUP; ++u) res[u] = res_next[u];
                    if (t0 + GROUP < NIT) { __builtin_amdgcn_sched_barrier(0); load_res(t0 + GROUP); __builtin_amdgcn_sched_barrier(0); }
                }
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
                    if constexpr (RES) {
                        const float4 x = res[u];
                        rr[u * CPL] = x.x; rr[u * CPL + 1] = x.y; rr[u * CPL + 2] = x.z; rr[u * CPL + 3] = x.w;
                        if (ACT == ACT_NONE && res_ln) {
                            const int r0 = (t0 + u) * RPI;
                            const float mean0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pst.x), r0));
                            const float mean1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pst.x), r0 + 1));
                            const float rstd0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pst.y), r0));
                            const float rstd1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pst.y), r0 + 1));
                            const float mean = rsub ? mean1 : mean0, rstd = rsub ? rstd1 : rstd0;
#pragma unroll
                            for (int c = 0; c < 4; ++c) rr[u * CPL + c] = ln_affine(rr[u * CPL + c], mean, rstd, lg[c], lb[c]);
                        }
                    }
                }
                if constexpr (FOLD) {
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
                if constexpr (LNP) {
                    // gemm4d's reduce-scatter over the 32 lanes of a row group (same operations in the same order: same
                    // bits); lane (idx & 7) == group keeps the totals of one row
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
                    float ks = b3 ? ks1 : ks0, kqq = b3 ? kq1 : kq0;
                    ks += g4d_xor_lane<8>(b3 ? ks0 : ks1); kqq += g4d_xor_lane<8>(b3 ? kq0 : kq1);
                    ks += g4d_xor_lane<4>(ks); kqq += g4d_xor_lane<4>(kqq);
                    ks += g4d_xor_lane<2>(ks); kqq += g4d_xor_lane<2>(kqq);
                    ks += g4d_xor_lane<1>(ks); kqq += g4d_xor_lane<1>(kqq);
                    const bool mine = (idx & 7) == (t0 >> 2);
                    acc_s = mine ? ks : acc_s; acc_q = mine ? kqq : acc_q;
                }
#pragma unroll
                for (int u = 0; u < GROUP; ++u) {
                    const int grow = grow0 + (t0 + u) * RPI;
                    if (!full && (grow >= g.M || !col_ok)) continue;
                    if constexpr (LO) {
                        const float4 a = make_float4(v[u * 8], v[u * 8 + 1], v[u * 8 + 2], v[u * 8 + 3]);
                        const float4 b = make_float4(v[u * 8 + 4], v[u * 8 + 5], v[u * 8 + 6], v[u * 8 + 7]);
                        store_out8<T>(e.out_lo + (size_t)grow * e.ld_lo + gcol, a, b);
                    } else {
                        float* d = e.out_f32 + (size_t)grow * e.ld_f32 + gcol;
                        const f32x4 x = {v[u * 4], v[u * 4 + 1], v[u * 4 + 2], v[u * 4 + 3]};
                        if (RES) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" :: "v"(d), "v"(x) : "memory");
                        else *(f32x4*)d = x;
                        if constexpr (EPI == G4D_EPI_BOTH || LNP)
                            store_out4<T>(e.out_lo + (size_t)grow * e.ld_lo + gcol, make_float4(x[0], x[1], x[2], x[3]));
                    }
                }
            }
            if constexpr (LNP) {
                // rows of a 32-row pass sit in the lanes whose (idx & 7) < 4
                const int rp = ((4 * (idx & 7) + 2 * ((idx >> 4) & 1) + ((idx >> 3) & 1)) << 1) + rsub;
                const int grow = prow0 + rp;
                if ((idx & 7) < 4 && grow < g.M && n0 + wn * 128 < g.N)
                    e.stats_part[(size_t)((n0 + wn * 128) >> 7) * e.ld_part + grow] = make_float2(acc_s, acc_q);
            }
        }
        range_report(e.range_flag, bad, W16 ? ZETT_RANGE_BIT_ACTIVATION : ZETT_RANGE_BIT_OUTPUT);
    }
}

template <typename T, int ACT, bool RES, int EPI>
inline hipError_t launch_gemm2w_inst(const GemmArgs<T>& g, hipStream_t stream) {
    static DeviceFlags attr;
    bool* done = attr.current();
    if (!done || !*done) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm2w_tn_kernel<T, ACT, RES, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, G2W_LDS_BYTES);
        if (e != hipSuccess) return e;
        if (done) *done = true;
    }
    const int tiles_m = (g.M + G2W_BM - 1) / G2W_BM;
    const int tiles_n = (g.N + G2W_BN - 1) / G2W_BN;
    if (tiles_m <= 0 || tiles_n <= 0) return hipSuccess;
    hipLaunchKernelGGL((gemm2w_tn_kernel<T, ACT, RES, EPI>), dim3(tiles_m * tiles_n), dim3(256), G2W_LDS_BYTES, stream, g);
    return hipGetLastError();
}

template <typename T, int ACT>
inline hipError_t launch_gemm2w_act(const GemmArgs<T>& g, hipStream_t stream, int mode) {
    const bool res = g.epi.residual != nullptr;
    switch (mode) {
        case G4D_EPI_LO: return launch_gemm2w_inst<T, ACT, false, G4D_EPI_LO>(g, stream);
        case G4D_EPI_LO_FOLD: return launch_gemm2w_inst<T, ACT, false, G4D_EPI_LO_FOLD>(g, stream);
        case G4D_EPI_F32:
            if constexpr (ACT == ACT_NONE) return res ? launch_gemm2w_inst<T, ACT, true, G4D_EPI_F32>(g, stream) : launch_gemm2w_inst<T, ACT, false, G4D_EPI_F32>(g, stream);
            else if constexpr (ACT == ACT_GELU_TANH) { if (res) return launch_gemm2w_inst<T, ACT, true, G4D_EPI_F32>(g, stream); }
            [[fallthrough]];
        default: return res ? launch_gemm2w_inst<T, ACT, true, G4D_EPI_GENERIC>(g, stream) : launch_gemm2w_inst<T, ACT, false, G4D_EPI_GENERIC>(g, stream);
    }
}

// Same epilogue selection as gemm4d (gemm4d_epi_mode): a launch gets the same arithmetic on either tile.
template <typename T>
inline hipError_t launch_gemm2w(const GemmArgs<T>& g, hipStream_t stream, bool force_generic = false) {
    if (g.K % 64 != 0) return hipErrorInvalidValue;
    const int mode = (force_generic && !g.epi.stats_part && !g.epi.fold_stats) ? G4D_EPI_GENERIC : gemm4d_epi_mode(g);
    if (mode < 0) return hipErrorInvalidValue;
    if (mode == G4D_EPI_F32_LN) return launch_gemm2w_inst<T, ACT_NONE, true, G4D_EPI_F32_LN>(g, stream);
    if (mode == G4D_EPI_F32_SCALE) return launch_gemm2w_inst<T, ACT_NONE, false, G4D_EPI_F32_SCALE>(g, stream);
    if (mode == G4D_EPI_BOTH) return launch_gemm2w_inst<T, ACT_NONE, false, G4D_EPI_BOTH>(g, stream);
    switch (g.epi.act) {
        case ACT_GELU_TANH: return launch_gemm2w_act<T, ACT_GELU_TANH>(g, stream, mode);
        case ACT_GELU_ERF: return launch_gemm2w_act<T, ACT_GELU_ERF>(g, stream, mode);
        default: return launch_gemm2w_act<T, ACT_NONE>(g, stream, mode);
    }
}

}  // namespace zett
