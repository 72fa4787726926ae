// gemm8x.hip.h — gemm8r.hip.h (256x256 tile, eight waves, register staging, one barrier per K step) on
// v_mfma_f32_16x16x32_{bf16,f16} instead of 32x32x16: the wave's 128x64 tile is 8x4 tiles of 16x16.
//
// Why: a register-only loop of the 16x16x32 shape sustains 2.2 PFLOP/s inside the chip's power envelope
// against 1.97 for 32x32x16 on the same random operands (tools/experiments/mfma_power.hip) — half the
// accumulator traffic per FLOP — and the GEMM is power-limited.  In the full kernel that is
// +2 % at K = 4096..8192 and +4 % at 8192^3, and -1..-4 % at K <= 2048 (more, shorter MFMA groups around the
// barrier), so the caller takes it for K >= 4096 without a residual epilogue.
//
// Both MFMA shapes accumulate the K products in ascending K order inside one fp32 chain, so this kernel
// returns the same bits as the 32x32x16 kernels (tile-variant test, fp32 outputs included).
//
// K step of a wave: 2 K blocks (32 wide) x 4 sub-blocks; a sub-block is 8 MFMAs on the A fragments of two
// 16-row tiles and the four W fragments of the K block.  A pairs are double-buffered, the W set of the other
// K block is read under the current one, the staging traffic of the next steps rides in sub-blocks 0-5,
// the barrier sits before the last two sub-blocks, which read the first fragments of the next step.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "gemm8r.hip.h"

namespace zett {

template <typename T> __device__ __forceinline__ f32x4 mfma16_kb(const u32x4& a, const u32x4& b, const f32x4& c);
template <> __device__ __forceinline__ f32x4 mfma16_kb<bf16_t>(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 mfma16_kb<f16_t>(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <typename T, int ACT = ACT_NONE, bool RES = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm8x_tn_kernel(GemmArgs<T> g) {
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
    constexpr int GROUP_M = ZETT_GROUP_M;
    const int group_size = GROUP_M * tiles_n;
    const int first_m = (wg / group_size) * GROUP_M;
    const int gm = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
    const int tm = first_m + (wg % group_size) % gm;
    const int tn = (wg % group_size) / gm;
    const int m0 = tm * G256_BM, n0 = tn * G256_BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // 0..7
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5;

    // staging plan: wave w moves rows w*32 + j*8 + lane/8 (j = 0..3) of each operand, 16-byte
    // chunk lane%8: uniform base pointer + 32-bit lane offset (rows past the edge are clamped)
    const unsigned char* a_base = (const unsigned char*)(g.A + (size_t)m0 * g.lda);
    const unsigned char* w_base = (const unsigned char*)(g.W + (size_t)n0 * g.ldw);
    uint32_t a_voff[4], w_voff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = wave * 32 + j * 8 + (lane >> 3);
        int ar = row; ar = m0 + ar < g.M ? ar : g.M - 1 - m0;
        int wr = row; wr = n0 + wr < g.N ? wr : g.N - 1 - n0;
        a_voff[j] = (uint32_t)ar * (uint32_t)g.lda * (uint32_t)sizeof(T) + (lane & 7) * 16;
        w_voff[j] = (uint32_t)wr * (uint32_t)g.ldw * (uint32_t)sizeof(T) + (lane & 7) * 16;
    }
    // ds_write address of piece j inside an operand image: row*128 + ((chunk ^ swz(row)) << 4);
    // swz(row) = (row>>1)&7 flips bit 2 between even and odd j
    int st_off[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int row = wave * 32 + par * 8 + (lane >> 3);
        st_off[par] = row * GEMM_ROW_BYTES + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
    }
    u32x4 ra[4], rw[4];
    // buffer loads: resource descriptor of the tile's operand panel in SGPRs, 32-bit lane offset,
    // K-step offset as the scalar offset operand -> no address arithmetic on the vector ALU
    // (global_load would re-add the step offset to sixteen 64-bit addresses per K step)
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, (short)0, 0x7fffffff, G4R_RSRC_WORD3);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)w_base, (short)0, 0x7fffffff, G4R_RSRC_WORD3);
    auto load_a = [&](int kt, int j) { ra[j] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_voff[j], kt * GEMM_ROW_BYTES, 0); };
    auto load_w = [&](int kt, int j) { rw[j] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_voff[j], kt * GEMM_ROW_BYTES, 0); };
    auto store_a = [&](int stage, int j) { *(u32x4*)(smem + stage * G256_STAGE_BYTES + st_off[j & 1] + (j >> 1) * 16 * GEMM_ROW_BYTES) = ra[j]; };
    auto store_w = [&](int stage, int j) { *(u32x4*)(smem + stage * G256_STAGE_BYTES + G256_OPERAND_BYTES + st_off[j & 1] + (j >> 1) * 16 * GEMM_ROW_BYTES) = rw[j]; };

    f32x4 acc[8][4];                 // 128x64 per wave as 8x4 tiles of 16x16
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
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
    // ---- prologue: step 0 through registers into stage 0, step 1 into registers
#pragma unroll
    for (int j = 0; j < 4; ++j) { load_a(0, j); load_w(0, j); }
#pragma unroll
    for (int j = 0; j < 4; ++j) { store_a(0, j); store_w(0, j); }
    if (nk > 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { load_a(1, j); load_w(1, j); }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(G4R_WAIT_LGKM0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) read_w(0, 0, j);
    read_a(0, 0, 0, 0); read_a(0, 0, 0, 1);

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
    int kt = 0;
    for (; kt + 2 < nk; ++kt) step(kt, yes_t{}, yes_t{});
    if (kt + 1 < nk) { step(kt, yes_t{}, no_t{}); ++kt; }
    step(kt, no_t{}, no_t{});

    // ---- epilogue as in gemm256.hip.h: 16 KiB region per wave (64 rows x 64 fp32), two passes
    __syncthreads();
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
        for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    region[(i4 * 16 + kq * 4 + r) * 64 + j * 16 + l15] = acc[4 * p + i4][j][r];
        if (RES || p == 0) __builtin_amdgcn_s_waitcnt(GEMM_WAIT_VMCNT0);
        Drain::drain(g, region, row0, gcol, col_ok, lane, bias8, sc8, sh8, oa, ob);
    }
}

template <typename T, int ACT, bool RES>
inline hipError_t launch_gemm8x_inst(const GemmArgs<T>& g, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm8x_tn_kernel<T, ACT, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int tiles_m = (g.M + G256_BM - 1) / G256_BM;
    const int tiles_n = (g.N + G256_BN - 1) / G256_BN;
    if (tiles_m <= 0 || tiles_n <= 0) return hipSuccess;
    hipLaunchKernelGGL((gemm8x_tn_kernel<T, ACT, RES>), dim3(tiles_m * tiles_n), dim3(512), G256_LDS_BYTES, stream, g);
    return hipGetLastError();
}

template <typename T, int ACT>
inline hipError_t launch_gemm8x_act(const GemmArgs<T>& g, hipStream_t stream) {
    return g.epi.residual ? launch_gemm8x_inst<T, ACT, true>(g, stream) : launch_gemm8x_inst<T, ACT, false>(g, stream);
}

template <typename T>
inline hipError_t launch_gemm8x(const GemmArgs<T>& g, hipStream_t stream) {
    switch (g.epi.act) {
        case ACT_GELU_TANH: return launch_gemm8x_act<T, ACT_GELU_TANH>(g, stream);
        case ACT_GELU_ERF: return launch_gemm8x_act<T, ACT_GELU_ERF>(g, stream);
        default: return launch_gemm8x_act<T, ACT_NONE>(g, stream);
    }
}

}  // namespace zett
