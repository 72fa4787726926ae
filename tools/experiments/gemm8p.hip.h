// gemm8p.hip.h — experiment: PERSISTENT form of gemm8x.hip.h (one workgroup per CU walks a tile list; the next
// tile's first K step is requested before the epilogue of the current one).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "gemm8x.hip.h"

namespace zett {

template <typename T, int ACT = ACT_NONE, bool RES = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm8p_tn_kernel(GemmArgs<T> g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BK = GEMM_ROW_BYTES / (int)sizeof(T);

    const int tiles_m = (g.M + G256_BM - 1) / G256_BM;
    const int tiles_n = (g.N + G256_BN - 1) / G256_BN;
    const int nwg = tiles_m * tiles_n;
    // persistent: XCD x (= blockIdx % 8) owns the contiguous tile range [x_first, x_first + x_count) of the grouped
    // order; its workgroups take every per_xcd-th tile of it
    const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int x_first = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int x_count = q8 + (xcd < r8 ? 1 : 0);
    constexpr int GROUP_M = 8;
    const int group_size = GROUP_M * tiles_n;
    int t = blockIdx.x >> 3;
    if (t >= x_count) return;
    int m0 = 0, n0 = 0;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // 0..7
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5;

    // staging plan of a tile: wave w moves rows w*32 + j*8 + lane/8 (j = 0..3) of each operand, 16-byte chunk lane%8
    uint32_t a_voff[4], w_voff[4];
    __amdgpu_buffer_rsrc_t a_rsrc, w_rsrc;
    auto set_tile = [&](int tt) {
        int lane_s = lane;
        asm volatile("" : "+v"(lane_s));                   // keeps the per-tile address arithmetic inside the tile loop
        const int wg = x_first + tt;
        const int first_m = (wg / group_size) * GROUP_M;
        const int gm = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
        m0 = (first_m + (wg % group_size) % gm) * G256_BM;
        n0 = ((wg % group_size) / gm) * G256_BN;
        a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(g.A + (size_t)m0 * g.lda), (short)0, 0x7fffffff, G4R_RSRC_WORD3);
        w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(g.W + (size_t)n0 * g.ldw), (short)0, 0x7fffffff, G4R_RSRC_WORD3);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = wave * 32 + j * 8 + (lane_s >> 3);
            const int ar = m0 + row < g.M ? row : g.M - 1 - m0;
            const int wr = n0 + row < g.N ? row : g.N - 1 - n0;
            a_voff[j] = (uint32_t)ar * (uint32_t)g.lda * (uint32_t)sizeof(T) + (lane_s & 7) * 16;
            w_voff[j] = (uint32_t)wr * (uint32_t)g.ldw * (uint32_t)sizeof(T) + (lane_s & 7) * 16;
        }
    };
    // ds_write address of piece j inside an operand image: row*128 + ((chunk ^ swz(row)) << 4);
    // swz(row) = (row>>1)&7 flips bit 2 between even and odd j
    int st_off[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int row = wave * 32 + par * 8 + (lane >> 3);
        st_off[par] = row * GEMM_ROW_BYTES + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
    }
    u32x4 ra[4], rw[4];
    auto load_a = [&](int kt, int j) { ra[j] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_voff[j], kt * GEMM_ROW_BYTES, 0); };
    auto load_w = [&](int kt, int j) { rw[j] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_voff[j], kt * GEMM_ROW_BYTES, 0); };
    auto store_a = [&](int stage, int j) { *(u32x4*)(smem + stage * G256_STAGE_BYTES + st_off[j & 1] + (j >> 1) * 16 * GEMM_ROW_BYTES) = ra[j]; };
    auto store_w = [&](int stage, int j) { *(u32x4*)(smem + stage * G256_STAGE_BYTES + G256_OPERAND_BYTES + st_off[j & 1] + (j >> 1) * 16 * GEMM_ROW_BYTES) = rw[j]; };

    f32x4 acc[8][4];                 // 128x64 per wave as 8x4 tiles of 16x16

    // fragment of a 16x16x32 MFMA: lane l holds row (l & 15), K elements (l >> 4)*8 .. +7 of a 32-wide K block,
    // i.e. 16-byte chunk kb*4 + (l >> 4) of the 128-byte row; 16-row steps leave the swizzle unchanged
    const int l15 = lane & 15, kq = lane >> 4;
    const int swz = (l15 >> 1) & 7;
    int a_off[2], w_off[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int c = ((kb * 4 + kq) ^ swz) << 4;
        a_off[kb] = (wm * 128 + l15) * GEMM_ROW_BYTES + c;
        w_off[kb] = G256_OPERAND_BYTES + (wn * 64 + l15) * GEMM_ROW_BYTES + c;
    }
    // a K step = 2 K blocks x 4 sub-blocks; sub-block sb uses the A fragments of row tiles 2s, 2s+1 (s = sb & 3) and the
    // four W fragments of its K block: 8 MFMAs.  A pairs are double-buffered, W sets alternate per K block.
    u32x4 fa[3][2], fw[2][4];     // A pair of sub-block sb lives in set sb % 3 (three sets: the pairs of the last two
                                  // sub-blocks are both read before the barrier)
    auto read_a = [&](int stage, int sb, int set, int h) {
        fa[set][h] = *(const u32x4*)(smem + stage * G256_STAGE_BYTES + a_off[sb >> 2] + (2 * (sb & 3) + h) * 16 * GEMM_ROW_BYTES);
    };
    auto read_w = [&](int stage, int kb, int j) {
        fw[kb][j] = *(const u32x4*)(smem + stage * G256_STAGE_BYTES + w_off[kb] + j * 16 * GEMM_ROW_BYTES);
    };
    auto mfma_one = [&](int sb, int m) {                     // m = 0..7: A fragment m >> 2 of the pair, W fragment m & 3
        const int i = 2 * (sb & 3) + (m >> 2), j = m & 3;
        acc[i][j] = mfma16_kb<T>(fa[sb % 3][m >> 2], fw[sb >> 2][j], acc[i][j]);
    };

    const int nk = g.K / BK;
    auto step = [&](int kt, auto more_c, auto more2_c) {
        constexpr bool more = decltype(more_c)::value, more2 = decltype(more2_c)::value;
        const int cur = kt & 1;
        // eight sub-blocks of 8 MFMAs, one MFMA per scheduling region.  Fillers: the A pair of the next sub-block
        // (2 reads), one W fragment of the other K block, and the staging traffic of the next steps.  The barrier
        // sits before sub-block 6: sub-blocks 6 and 7 (16 MFMAs) cover the first fragment reads of step kt+1.
#pragma unroll
        for (int sb = 0; sb < 8; ++sb) {
            if (sb == 6) {
                // every read of stage cur and every write of stage cur^1 by this wave is complete
                __builtin_amdgcn_s_waitcnt(G4R_WAIT_LGKM0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                mfma_one(sb, m);
                if (sb < 5) {
                    if (m == 0) read_a(cur, sb + 1, (sb + 1) % 3, 0);
                    if (m == 1) read_a(cur, sb + 1, (sb + 1) % 3, 1);
                    if (sb < 4 && m == 2) read_w(cur, 1, sb);                        // W fragments of K block 1
                    if (sb == 0 && (m == 4 || m == 6) && more) store_a(cur ^ 1, (m - 4) >> 1);
                    if (sb == 1 && (m == 4 || m == 6) && more) store_a(cur ^ 1, 2 + ((m - 4) >> 1));
                    if (sb == 1 && (m == 5 || m == 7) && more2) load_a(kt + 2, (m - 5) >> 1);
                    if (sb == 2 && (m == 5 || m == 7) && more2) load_a(kt + 2, 2 + ((m - 5) >> 1));
                    if (sb == 2 && (m == 4 || m == 6) && more) store_w(cur ^ 1, (m - 4) >> 1);
                    if (sb == 3 && (m == 4 || m == 6) && more) store_w(cur ^ 1, 2 + ((m - 4) >> 1));
                    if (sb == 4 && (m == 4 || m == 6) && more2) load_w(kt + 2, (m - 4) >> 1);
                } else if (sb == 5) {                               // pairs of sub-blocks 6 and 7
                    if (m == 0) read_a(cur, 6, 0, 0);
                    if (m == 1) read_a(cur, 6, 0, 1);
                    if (m == 2) read_a(cur, 7, 1, 0);
                    if (m == 3) read_a(cur, 7, 1, 1);
                    if ((m == 4 || m == 6) && more2) load_w(kt + 2, 2 + ((m - 4) >> 1));
                } else if (more) {                                  // first fragments of step kt+1
                    if (sb == 6 && m < 4) read_w(cur ^ 1, 0, m);
                    if (sb == 7 && m == 0) read_a(cur ^ 1, 0, 0, 0);
                    if (sb == 7 && m == 1) read_a(cur ^ 1, 0, 0, 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    typedef std::integral_constant<bool, true> yes_t;
    typedef std::integral_constant<bool, false> no_t;
    typedef EpiDrain<T, ACT, RES, 64, 64> Drain;
    float* region = (float*)(smem + wave * 16384);

    set_tile(t);
#pragma unroll
    for (int j = 0; j < 4; ++j) { load_a(0, j); load_w(0, j); }          // step 0 of the first tile
    for (; t < x_count; t += per_xcd) {
        // ---- tile prologue: step 0 (requested under the previous epilogue) through stage 0, step 1 into registers
#pragma unroll
        for (int j = 0; j < 4; ++j) { store_a(0, j); store_w(0, j); }
        if (nk > 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { load_a(1, j); load_w(1, j); }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(G4R_WAIT_LGKM0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) read_w(0, 0, j);
        read_a(0, 0, 0, 0); read_a(0, 0, 0, 1);

        int kt = 0;
        for (; kt + 2 < nk; ++kt) step(kt, yes_t{}, yes_t{});
        if (kt + 1 < nk) { step(kt, yes_t{}, no_t{}); ++kt; }
        step(kt, no_t{}, no_t{});

        // ---- tile boundary: every wave is past its last fragment reads after this barrier (the stages become the
        // epilogue regions); the staging registers are free, so the next tile's first K step is requested now
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(G4R_WAIT_LGKM0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const int em0 = m0, en0 = n0;
        if (t + per_xcd < x_count) {
            set_tile(t + per_xcd);
#pragma unroll
            for (int j = 0; j < 4; ++j) { load_a(0, j); load_w(0, j); }
        }
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int gcol = en0 + wn * 64 + (lane_e % Drain::LPR) * 8;
        const bool col_ok = gcol < g.N;
        float4 bias8[2], sc8[2], sh8[2];
        Drain::load_cols(g.epi, gcol, col_ok, bias8, sc8, sh8);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float4 oa[Drain::NIT], ob[Drain::NIT];
            const int row0 = em0 + wm * 128 + p * 64;
            Drain::load_res(g, row0, gcol, col_ok, lane_e, oa, ob);
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        region[(i4 * 16 + kq * 4 + r) * 64 + j * 16 + l15] = acc[4 * p + i4][j][r];
            // NOTE: this also waits for the next tile's step-0 loads (they are older than nothing the drain needs)
            if (RES || p == 0) __builtin_amdgcn_s_waitcnt(GEMM_WAIT_VMCNT0);
            Drain::drain(g, region, row0, gcol, col_ok, lane_e, bias8, sc8, sh8, oa, ob);
        }
        // every wave is done with its region before the next tile's ds_writes reuse the stages
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(G4R_WAIT_LGKM0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <typename T, int ACT, bool RES>
inline hipError_t launch_gemm8p_inst(const GemmArgs<T>& g, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm8p_tn_kernel<T, ACT, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int tiles_m = (g.M + G256_BM - 1) / G256_BM;
    const int tiles_n = (g.N + G256_BN - 1) / G256_BN;
    if (tiles_m <= 0 || tiles_n <= 0) return hipSuccess;
    const int nwg = tiles_m * tiles_n;
    const int grid = nwg >= 256 ? 256 : ((nwg + 7) / 8) * 8;        // one workgroup per CU, a multiple of the 8 XCDs
    hipLaunchKernelGGL((gemm8p_tn_kernel<T, ACT, RES>), dim3(grid), dim3(512), G256_LDS_BYTES, stream, g);
    return hipGetLastError();
}

template <typename T, int ACT>
inline hipError_t launch_gemm8p_act(const GemmArgs<T>& g, hipStream_t stream) {
    return g.epi.residual ? launch_gemm8p_inst<T, ACT, true>(g, stream) : launch_gemm8p_inst<T, ACT, false>(g, stream);
}

template <typename T>
inline hipError_t launch_gemm8p(const GemmArgs<T>& g, hipStream_t stream) {
    switch (g.epi.act) {
        case ACT_GELU_TANH: return launch_gemm8p_act<T, ACT_GELU_TANH>(g, stream);
        case ACT_GELU_ERF: return launch_gemm8p_act<T, ACT_GELU_ERF>(g, stream);
        default: return launch_gemm8p_act<T, ACT_NONE>(g, stream);
    }
}

}  // namespace zett
