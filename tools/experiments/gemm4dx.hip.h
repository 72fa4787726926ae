// gemm4dx.hip.h — EXPERIMENT (tools/gemm_bench only, not in the library): 256x256 tile, FOUR waves (one per
// SIMD, 128x128 of the tile each), both operands streamed HBM/L2 -> LDS by buffer_load_dwordx4 ... lds
// (no VGPR round trip, no ds_write pass), on v_mfma_f32_16x16x32_{bf16,f16}, accumulators pinned to AGPRs.
// This is the geometry of the hipBLASLt kernel the yardstick runs (MT256x256x64, MI16x16, 256 threads,
// direct-to-LDS on both operands), rebuilt from its visible schedule.
//
//   C[M,N] = epilogue( A[M,K] · W[N,K]ᵀ )      (contract and epilogue of gemm.hip.h; bit-identical results)
//
//   K step t of a wave = 128 MFMAs, K block 0 (64) then K block 1 (64); stage t&1 holds step t.
//     under block 0:  read the W fragments of block 1              -> wait, barrier B1: W image free
//                     request W rows of step t+2 into the same stage, read the A fragments of block 1
//                                                                  -> wait, barrier B2: A image free
//                     request A rows of step t+2 (continues under block 1)
//     under block 1:  vmcnt(24) + barrier B3: W of step t+1 has landed for every wave
//                     read the W fragments of step t+1, block 0
//                     vmcnt(16) + barrier B4: A of step t+1 has landed
//                     read the A fragments of step t+1, block 0
//   The 16 requests of a wave and step are spread one per 2-4 MFMAs (the compiler keeps the order: one
//   MFMA per scheduling region), fragments are double-buffered in 128 VGPRs next to 256 AGPRs.
//
// Measured on MI355X (tools/gemm_bench, BURST=20, bf16 out; TFLOP/s):
//                                   g8r    g8x    g4d    hipBLASLt
//   8192^3, randn operands          1379   1446   1427   1594
//   65536x4096x4096, randn          1350   1373   1333
//   65536x1024x4096, randn          1305   1302   1360   1466
//   65536x4096x4096, fp32 residual  1101   1091   1129
//   8192^3, ALL-ZERO operands       1945   1848   1519   2259     (no power limit: the schedule alone)
//   8192^3, zero, lda = 0 (all hit) 2070   1948   1986
//   ablations at 8192^3, zero:      no DMA 2158 | DMA but no barriers/vmcnt waits 2073 | MFMAs only 2206
// Reading: the issue cost of the spread requests is small (2158 -> 2073), the schedule is sound (1986 when
// every request hits), and what costs 30 % is WAITING for the data: LDS holds two K steps, so a request has
// 1.2-1.4 steps to land, less than an L2 miss takes under load; register staging (gemm8r) keeps a third
// step in VGPRs and loses only 6 % to the same misses.  A software prefetch of step t+4 into L2 (PF > 0:
// one junk dword per line) makes it worse (1224): loads complete in order, so the slow touches sit in
// front of the requests in the vmcnt queue.  Part of that latency is channel camping: every tile walks K
// from 0, and with row strides of 8-16 KiB all tiles read the same 256-byte columns at the same time.
// Starting the K loop of tile (tm, tn) at step ((tm + tn) % 32) * 2 and wrapping (PF = 101, Tensile's
// "StaggerU") gives 1745 on zero operands (+13 %) and 1461 / 1364 on randn operands (+3 %); staggering by
// tn alone (PF = 102), the only form that keeps a row's summation order independent of where the row sits
// in the batch (the invariant tests/test_invariants_gpu.py holds every tile variant to), gives nothing:
// the tiles an XCD runs together differ mostly in tm.  Under real operands the chip is power-limited and
// the three kernels end within 3 % of each other, so the library keeps the register-staged pair.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "gemm4d.hip.h"

#ifndef G4DX_AUX
#define G4DX_AUX 0      // cache-policy bits of the LDS-DMA requests (gfx940: 1 = sc0, 2 = nt, 16 = sc1)
#endif
#ifndef G4D_ABL
#define G4D_ABL 0      // ablation bits for tools/gemm_bench (results are garbage when set): 1 no DMA in the loop,
#endif                 // 2 no B1/B2, 4 no B3/B4 + vmcnt waits, 8 no fragment reads in the loop, 16 no s_barrier (waits kept), 32 no vmcnt waits (barriers kept)

namespace zett {

__device__ int g4dx_cfg[2] = {4, 0};     // GROUP_M, map mode of the PF = 230 variant

// PF > 0: every K step also touches one dword of each 128-byte operand line of step t+PF (one load per wave
// and operand into a junk register that is never read): the lines are in L2 when their LDS-DMA request is
// issued PF-2 steps later.  LDS holds two steps, so a request has one step to land - less than a miss to
// HBM/MALL takes under load (all-hit addressing runs 30 % faster without this).
template <typename T, int ACT = ACT_NONE, bool RES = false, int PF = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm4dx_tn_kernel(GemmArgs<T> g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BK = GEMM_ROW_BYTES / (int)sizeof(T);

    const int tiles_m = (g.M + G256_BM - 1) / G256_BM;
    const int tiles_n = (g.N + G256_BN - 1) / G256_BN;
    const int nwg = tiles_m * tiles_n;
    int wg = blockIdx.x;
    // PF = 230: tile order read at run time from g4dx_cfg (tools/gemm_bench G4DX_GROUP_M / G4DX_MAP): the sweep behind
    // the library's ZETT_GROUP_M.  map 0 = XCD chunks (each XCD a contiguous range of the order), 1 = no remap (the
    // hardware's round-robin: neighbouring tiles of the order on different XCDs), 2 = XCD chunks, N-major groups.
    const int map_mode = PF == 230 ? g4dx_cfg[1] : 0;
    if (map_mode != 1 && map_mode != 3) {
        const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
    }
    const int GROUP_M = PF == 230 ? g4dx_cfg[0] : (PF == 212 || PF == 213) ? 4 : PF == 214 ? 2 : PF == 215 ? 1 : PF == 216 ? 6 : 8;
    int tm, tn;
    if (map_mode >= 2) {          // (3 = as 2 without the XCD remap) groups of GROUP_M column tiles, row tiles outermost inside a group
        const int group_size = GROUP_M * tiles_m;
        const int first_n = (wg / group_size) * GROUP_M;
        const int gn = (tiles_n - first_n) < GROUP_M ? (tiles_n - first_n) : GROUP_M;
        tn = first_n + (wg % group_size) % gn;
        tm = (wg % group_size) / gn;
    } else {
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
    const int nk = g.K / BK;
    // PF = 101 / 102: the K loop of a tile starts at step ((tm + tn) / tn only) % 32 * 2 and wraps (what Tensile calls
    // StaggerU): tiles running at the same time then read different 256-byte columns, i.e. different memory channels.
    // 101 makes the summation order depend on the row tile and is for measurement only.
    const int stag = (PF == 101 || PF == 201 || PF == 211) ? (((tm + tn) & 31) * 2) % nk : (PF == 102 || PF == 212) ? ((tn & 31) * 2) % nk : 0;
    uint32_t a_touch, w_touch, junk = 0;
    {
        const int row = wave * 64 + lane;
        int ar = row; ar = m0 + ar < g.M ? ar : g.M - 1 - m0;
        int wr = row; wr = n0 + wr < g.N ? wr : g.N - 1 - n0;
        a_touch = (uint32_t)ar * (uint32_t)g.lda * (uint32_t)sizeof(T);
        w_touch = (uint32_t)wr * (uint32_t)g.ldw * (uint32_t)sizeof(T);
    }
    auto touch = [&](const unsigned char* base, uint32_t voff, int kt) {
        const unsigned char* src = base + (size_t)kt * GEMM_ROW_BYTES;
        asm volatile("global_load_dword %0, %1, %2" : "+v"(junk) : "v"(voff), "s"(src));
    };
    u32x4 junk4 = {0, 0, 0, 0};       // G4D_ABL & 64: the requests fetch into this register instead of LDS
    auto dma_a = [&](int kt, int r) {
        int ks = kt + stag; if (ks >= nk) ks -= nk;
        if (G4D_ABL & 64) { asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(junk4) : "v"(a_voff[r]), "s"(a_rsrc), "s"(ks * GEMM_ROW_BYTES)); return; }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_ptr_t)(my_rows + (kt & 1) * G256_STAGE_BYTES + r * 8 * GEMM_ROW_BYTES), 16,
                                                 a_voff[r], ks * GEMM_ROW_BYTES, 0, G4DX_AUX);
    };
    auto dma_w = [&](int kt, int r) {
        int ks = kt + stag; if (ks >= nk) ks -= nk;
        if (G4D_ABL & 64) { asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(junk4) : "v"(w_voff[r]), "s"(w_rsrc), "s"(ks * GEMM_ROW_BYTES)); return; }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(my_rows + (kt & 1) * G256_STAGE_BYTES + G256_OPERAND_BYTES + r * 8 * GEMM_ROW_BYTES), 16,
                                                 w_voff[r], ks * GEMM_ROW_BYTES, 0, G4DX_AUX);
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
    const int l15 = lane & 15, kq = lane >> 4;
    const int swz = (l15 >> 1) & 7;
    int a_off[2], w_off[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int c = ((kb * 4 + kq) ^ swz) << 4;
        a_off[kb] = (wm * 128 + l15) * GEMM_ROW_BYTES + c;
        w_off[kb] = G256_OPERAND_BYTES + (wn * 128 + l15) * GEMM_ROW_BYTES + c;
    }
    u32x4 fa[2][8], fw[2][8];
    auto read_a = [&](int stage, int kb, int i) {
        fa[kb][i] = *(const u32x4*)(smem + stage * G256_STAGE_BYTES + a_off[kb] + i * 16 * GEMM_ROW_BYTES);
    };
    auto read_w = [&](int stage, int kb, int j) {
        fw[kb][j] = *(const u32x4*)(smem + stage * G256_STAGE_BYTES + w_off[kb] + j * 16 * GEMM_ROW_BYTES);
    };

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

    // more: step kt+1 exists (its block-0 fragments are read here); more2: step kt+2 exists (requested here)
    auto step = [&](int kt, auto more_c, auto more2_c, auto wave_c) {
        constexpr bool more = decltype(more_c)::value, more2 = decltype(more2_c)::value;
        constexpr int WV = decltype(wave_c)::value;     // schedule 2: this copy of the loop belongs to wave WV
        const int cur = kt & 1;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int p = kb * 64 + i * 8 + j;
            if constexpr (PF == 240) {
                // schedule 5: ONE barrier per step (p = 102): it both admits the block-0 reads of step kt+1 (vmcnt) and
                // releases stage cur (every wave read its last fragment of it before p = 31).  The 16 requests of step
                // kt+2 follow at p = 103..126 (12 slots x 2 waves) and, in the next step, p = 0..39: half a step to land.
                // Measured: correct, and 7-13 % SLOWER than schedule 3 on real operands (1 607 vs 1 984 TFLOP/s on zeros):
                // with half a step of lookahead the vmcnt wait does stall.  A request needs about one K step to land;
                // the number of barriers (1, 2, 3 or 4 per step) is not what separates these schedules.
                if (p == 102 && more) {
                    __builtin_amdgcn_s_waitcnt(g4d_wait_vm(0));
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
                mfma16_agpr<T>(acc[i][j], fa[kb][i], fw[kb][j]);
                if (p < 16 && (p & 1) == 0) read_w(cur, 1, p >> 1);
                if (p >= 16 && p <= 30 && (p & 1) == 0) read_a(cur, 1, (p - 16) >> 1);
                // requests of step kt+2 (issued in step kt after the barrier) and of step kt+1 (the rest, issued here)
                if (more2 && p >= 103 && p < 127 && WV == ((p - 103) & 3)) dma_w(kt + 2, (p - 103) >> 2);          // W 0..5
                if (more && p < 40 && WV == (p & 3)) {
                    const int r = p >> 2;                                                                   // 0..9
                    if (r < 2) dma_w(kt + 1, 6 + r); else dma_a(kt + 1, r - 2);
                }
                if (more && p >= 103 && p <= 110) read_w(cur ^ 1, 0, p - 103);
                if (more && p >= 111 && p <= 125 && (p & 1) == 1) read_a(cur ^ 1, 0, (p - 111) >> 1);
                __builtin_amdgcn_sched_barrier(0);
                continue;
            }
            if constexpr (PF >= 220) {
                // schedule 4: as 3 with the block-1 reads one per MFMA (p = 0..15), 12 MFMAs of slack before the release
                // barrier (p = 28), requests over p = 30..93, landed barrier at p = 96, 31 slots for the 16 block-0 reads
                if (p == 28 && more2) {
                    __builtin_amdgcn_s_waitcnt(G4R_WAIT_LGKM0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (p == 96 && more) {
                    __builtin_amdgcn_s_waitcnt(g4d_wait_vm(more2 ? 16 : 0));
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
                mfma16_agpr<T>(acc[i][j], fa[kb][i], fw[kb][j]);
                if (p < 8) read_w(cur, 1, p);
                if (p >= 8 && p < 16) read_a(cur, 1, p - 8);
                if (more2 && p >= 30 && p < 62 && WV == ((p - 30) & 3)) dma_w(kt + 2, (p - 30) >> 2);
                if (more2 && p >= 62 && p < 94 && WV == ((p - 62) & 3)) dma_a(kt + 2, (p - 62) >> 2);
                if (more && p >= 97 && p <= 111 && (p & 1) == 1) read_w(cur ^ 1, 0, (p - 97) >> 1);
                if (more && p >= 112 && p <= 126 && (p & 1) == 0) read_a(cur ^ 1, 0, (p - 112) >> 1);
                __builtin_amdgcn_sched_barrier(0);
                continue;
            }
            if constexpr (PF >= 210) {
                // schedule 3: two barriers per step.  All block-1 fragments are read under MFMAs 0..30, one barrier
                // releases both images of stage cur (p = 36), the 16 requests of step kt+2 are paced over p = 38..101,
                // one wait + barrier (p = 102) admits the block-0 reads of step kt+1.
                if (p == 36 && more2) {
                    __builtin_amdgcn_s_waitcnt(G4R_WAIT_LGKM0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (p == 102 && more) {
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
                continue;
            }
            if constexpr (PF >= 200) {
                // schedule 2: three barriers per step, and the four waves take turns at the texture path - wave w
                // issues its request r of an operand under MFMA base + 4r + w, so that the CU sees one request
                // per MFMA instead of four at once every fourth
                if ((p == 18 || p == 40) && more2) {
                    __builtin_amdgcn_s_waitcnt(G4R_WAIT_LGKM0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (p == 84 && more) {              // everything requested during the previous step has landed
                    __builtin_amdgcn_s_waitcnt(g4d_wait_vm(more2 ? 16 : 0));
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
                mfma16_agpr<T>(acc[i][j], fa[kb][i], fw[kb][j]);
                if (p < 16 && (p & 1) == 0) read_w(cur, 1, p >> 1);
                if (p >= 21 && p <= 35 && (p & 1) == 1) read_a(cur, 1, (p - 21) >> 1);
                if (more2 && p >= 20 && p < 52 && WV == ((p - 20) & 3)) dma_w(kt + 2, (p - 20) >> 2);
                if (more2 && p >= 52 && p < 84 && WV == ((p - 52) & 3)) dma_a(kt + 2, (p - 52) >> 2);
                if (more && p >= 85 && p <= 99 && (p & 1) == 1) read_w(cur ^ 1, 0, (p - 85) >> 1);
                if (more && p >= 100 && p <= 121 && (p - 100) % 3 == 0) read_a(cur ^ 1, 0, (p - 100) / 3);
                __builtin_amdgcn_sched_barrier(0);
                continue;
            }
            if (p == 18 || p == 40) {
                // this wave's reads of the W (p = 18) / A (p = 40) image of stage cur are complete
                if (more2 && !(G4D_ABL & 2)) {
                    __builtin_amdgcn_s_waitcnt(G4R_WAIT_LGKM0);
                    if (!(G4D_ABL & 16)) __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (p == 72 && more && !(G4D_ABL & 4)) {          // W of step kt+1: everything requested after it may still be in flight
                if (!(G4D_ABL & 32)) __builtin_amdgcn_s_waitcnt(g4d_wait_vm(more2 ? (PF && PF < 100 ? 26 : 24) : 8));
                if (!(G4D_ABL & 16)) __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            if (p == 96 && more && !(G4D_ABL & 4)) {          // A of step kt+1
                if (!(G4D_ABL & 32)) __builtin_amdgcn_s_waitcnt(g4d_wait_vm(more2 ? (PF && PF < 100 ? 18 : 16) : 0));
                if (!(G4D_ABL & 16)) __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            mfma16_agpr<T>(acc[i][j], fa[kb][i], fw[kb][j]);
            if (PF && PF < 100 && more2 && p == 4) touch(a_base, a_touch, kt + PF < nk ? kt + PF : nk - 1);
            if (PF && PF < 100 && more2 && p == 12) touch(w_base, w_touch, kt + PF < nk ? kt + PF : nk - 1);
            // block-1 fragments of this step
            if (!(G4D_ABL & 8) && p < 16 && (p & 1) == 0) read_w(cur, 1, p >> 1);
            if (!(G4D_ABL & 8) && p >= 21 && p <= 35 && (p & 1) == 1) read_a(cur, 1, (p - 21) >> 1);
            // requests of step kt+2 into the images just released
            if (!(G4D_ABL & 1) && more2 && p >= 20 && p <= 34 && (p & 1) == 0) dma_w(kt + 2, (p - 20) >> 1);
            if (!(G4D_ABL & 1) && more2 && p >= 42 && p <= 70 && ((p - 42) & 3) == 0) dma_a(kt + 2, (p - 42) >> 2);
            // block-0 fragments of step kt+1
            if (!(G4D_ABL & 8) && more && p >= 74 && p <= 88 && (p & 1) == 0) read_w(cur ^ 1, 0, (p - 74) >> 1);
            if (!(G4D_ABL & 8) && more && p >= 98 && p <= 112 && (p & 1) == 0) read_a(cur ^ 1, 0, (p - 98) >> 1);
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
    if constexpr (PF >= 200) {          // one copy of the loop per wave: the request slots differ
        if (wave == 0) k_loop(std::integral_constant<int, 0>{});
        else if (wave == 1) k_loop(std::integral_constant<int, 1>{});
        else if (wave == 2) k_loop(std::integral_constant<int, 2>{});
        else k_loop(std::integral_constant<int, 3>{});
    } else {
        k_loop(std::integral_constant<int, 0>{});
    }

    // ---- epilogue: each wave stages its 128x128 quadrant through a private 32 KiB LDS region
    // (64 rows x 128 fp32), two passes, drained by EpiDrain (gemm256.hip.h).
    asm volatile("s_nop 15\n\ts_nop 15");     // last MFMA (8 passes) -> first accumulator read
    if (PF) { __builtin_amdgcn_s_waitcnt(GEMM_WAIT_VMCNT0); asm volatile("" :: "v"(junk)); }
    if (G4D_ABL & 64) { __builtin_amdgcn_s_waitcnt(GEMM_WAIT_VMCNT0); asm volatile("" :: "v"(junk4)); }
    __syncthreads();
    float* region = (float*)(smem + wave * 32768);
    typedef EpiDrain<T, ACT, RES, 64, 128, true, false> Drain;
    const int gcol = n0 + wn * 128 + (lane % Drain::LPR) * 8;
    const bool col_ok = gcol < g.N;
    float4 bias8[2], sc8[2], sh8[2];
    Drain::load_cols(g.epi, gcol, col_ok, bias8, sc8, sh8);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        float4 oa[Drain::NIT], ob[Drain::NIT];
        const int row0 = m0 + wm * 128 + p * 64;
        Drain::load_res(g, row0, gcol, col_ok, lane, oa, ob);
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    region[(i4 * 16 + kq * 4 + r) * 128 + j * 16 + l15] = acc[4 * p + i4][j][r];
        if (RES || p == 0) __builtin_amdgcn_s_waitcnt(GEMM_WAIT_VMCNT0);
        Drain::drain(g, region, row0, gcol, col_ok, lane, bias8, sc8, sh8, oa, ob);
    }
}

template <typename T, int ACT, bool RES, int PF>
inline hipError_t launch_gemm4dx_inst(const GemmArgs<T>& g, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm4dx_tn_kernel<T, ACT, RES, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int tiles_m = (g.M + G256_BM - 1) / G256_BM;
    const int tiles_n = (g.N + G256_BN - 1) / G256_BN;
    if (tiles_m <= 0 || tiles_n <= 0) return hipSuccess;
    hipLaunchKernelGGL((gemm4dx_tn_kernel<T, ACT, RES, PF>), dim3(tiles_m * tiles_n), dim3(256), G256_LDS_BYTES, stream, g);
    return hipGetLastError();
}

template <typename T, int ACT, int PF>
inline hipError_t launch_gemm4dx_act(const GemmArgs<T>& g, hipStream_t stream) {
    return g.epi.residual ? launch_gemm4dx_inst<T, ACT, true, PF>(g, stream) : launch_gemm4dx_inst<T, ACT, false, PF>(g, stream);
}

template <typename T, int PF = 0>
inline hipError_t launch_gemm4dx(const GemmArgs<T>& g, hipStream_t stream) {
    switch (g.epi.act) {
        case ACT_GELU_TANH: return launch_gemm4dx_act<T, ACT_GELU_TANH, PF>(g, stream);
        case ACT_GELU_ERF: return launch_gemm4dx_act<T, ACT_GELU_ERF, PF>(g, stream);
        default: return launch_gemm4dx_act<T, ACT_NONE, PF>(g, stream);
    }
}

}  // namespace zett
