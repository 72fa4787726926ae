// gemm8r.hip.h — the default 256x256 MFMA GEMM for gfx950: eight waves (2(M) x 4(N), 128x64 each, two
// per SIMD) with REGISTER staging — buffer_load_dwordx4 -> VGPR -> ds_write_b128 — and one barrier
// per K step.
//
//   C[M,N] = epilogue( A[M,K] · W[N,K]ᵀ )      (contract and epilogue of gemm.hip.h)
//
// It is gemm4r.hip.h's pipeline (see there for why LDS-DMA lost: a 1 KiB LDS-DMA request blocks the
// issue port of its SIMD for ~52 cycles) on gemm256.hip.h's wave geometry: a K step of a wave is 32
// MFMAs in four groups of 8, one MFMA per scheduling region; group 0 writes the wave's four A
// pieces of step t+1 (in registers since step t-1) to the other LDS stage and reloads those
// registers with step t+2, group 1 does the same for W, the barrier sits between group 2 and 3,
// group 3 reads the first fragments of step t+1.  32 staging registers per lane (246 VGPRs in
// all, two waves per SIMD).  With two waves per SIMD the second wave covers what the first cannot
// hide, and prologue and epilogue are those of the eight-wave kernel: it beats the LDS-DMA kernel
// by 7-11 % at every K (K = 768: 898 -> 980, K = 4096: 1182 -> 1313 TFLOP/s) and the four-wave
// kernel by 2-12 % (tools/gemm_bench).  Same K reduction order and epilogue
// arithmetic as every other tile variant: identical bits.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "gemm_tile.hip.h"

namespace zett {

template <typename T, int ACT = ACT_NONE, bool RES = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm8r_tn_kernel(GemmArgs<T> g) {
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

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int swz = (l31 >> 1) & 7;
    int a_off[4], w_off[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int c = ((kk * 2 + hi) ^ swz) << 4;
        a_off[kk] = (wm * 128 + l31) * GEMM_ROW_BYTES + c;
        w_off[kk] = G256_OPERAND_BYTES + (wn * 64 + l31) * GEMM_ROW_BYTES + c;
    }
    u32x4 fa[2][4], fw[2][2];
    auto read_frag = [&](int stage, int kk, int set, int q) {     // q = 0..3: A fragment q, 4..5: W fragment q-4
        const unsigned char* S = smem + stage * G256_STAGE_BYTES;
        if (q < 4) fa[set][q] = *(const u32x4*)(S + a_off[kk] + q * 32 * GEMM_ROW_BYTES);
        else fw[set][q - 4] = *(const u32x4*)(S + w_off[kk] + (q - 4) * 32 * GEMM_ROW_BYTES);
    };
    auto mfma_one = [&](int set, int m) { mfma_chunk<T>(fa[set][m >> 1], fw[set][m & 1], acc[m >> 1][m & 1]); };

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
    for (int q = 0; q < 6; ++q) read_frag(0, 0, 0, q);

    auto step = [&](int kt, auto more_c, auto more2_c) {
        constexpr bool more = decltype(more_c)::value, more2 = decltype(more2_c)::value;
        const int cur = kt & 1;
        // group 0 (8 MFMAs, one per scheduling region): fragments of kk=1; A rows of step kt+1 -> stage cur^1,
        // reload with step kt+2
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            mfma_one(0, q);
            if (q < 6) read_frag(cur, 1, 1, q);
            if ((q & 1) && more) store_a(cur ^ 1, q >> 1);
            if ((q & 1) && more2) load_a(kt + 2, q >> 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        // group 1: fragments of kk=2; W rows of step kt+1 -> stage cur^1
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            mfma_one(1, q);
            if (q < 6) read_frag(cur, 2, 0, q);
            if ((q & 1) && more) store_w(cur ^ 1, q >> 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        // group 2: fragments of kk=3; the W staging registers are reloaded with step kt+2
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            mfma_one(0, q);
            if (q < 6) read_frag(cur, 3, 1, q);
            if ((q & 1) && more2) load_w(kt + 2, q >> 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        // every read of stage cur and every write of stage cur^1 by this wave is complete
        __builtin_amdgcn_s_waitcnt(G4R_WAIT_LGKM0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // group 3: first fragments of step kt+1
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            mfma_one(1, q);
            if (q < 6 && more) read_frag(cur ^ 1, 0, 0, q);
            __builtin_amdgcn_sched_barrier(0);
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
    typedef EpiDrain<T, ACT, RES, 64, 64, !RES> Drain;      // (no Rescaler behind a residual: the launcher refuses the pair)
    const int gcol = n0 + wn * 64 + (lane % Drain::LPR) * 8;
    const bool col_ok = gcol < g.N;
    float4 bias8[2], sc8[2], sh8[2];
    Drain::load_cols(g.epi, gcol, col_ok, bias8, sc8, sh8);
    float4 lng[2], lnb[2];
    Drain::load_ln_cols(g.epi, gcol, col_ok, lng, lnb);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        float4 oa[Drain::NIT], ob[Drain::NIT];
        const int row0 = m0 + wm * 128 + p * 64;
        Drain::load_res(g, row0, gcol, col_ok, lane, oa, ob);
        float2 lnst;
        Drain::load_res_stats(g, row0, lane, lnst);
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    region[(i2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * 64 + j * 32 + l31] = acc[2 * p + i2][j][r];
        if (RES || p == 0) __builtin_amdgcn_s_waitcnt(GEMM_WAIT_VMCNT0);
        Drain::ln_res(g, lane, lng, lnb, lnst, oa, ob);
        Drain::drain(g, region, row0, gcol, col_ok, lane, bias8, sc8, sh8, oa, ob);
    }
}

template <typename T, int ACT, bool RES>
inline hipError_t launch_gemm8r_inst(const GemmArgs<T>& g, hipStream_t stream) {
    static DeviceFlags attr;      // per device: one process may hold a handle per GPU
    bool* done = attr.current();
    if (!done || !*done) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm8r_tn_kernel<T, ACT, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS_BYTES);
        if (e != hipSuccess) return e;
        if (done) *done = true;
    }
    const int tiles_m = (g.M + G256_BM - 1) / G256_BM;
    const int tiles_n = (g.N + G256_BN - 1) / G256_BN;
    if (tiles_m <= 0 || tiles_n <= 0) return hipSuccess;
    hipLaunchKernelGGL((gemm8r_tn_kernel<T, ACT, RES>), dim3(tiles_m * tiles_n), dim3(512), G256_LDS_BYTES, stream, g);
    return hipGetLastError();
}

template <typename T, int ACT>
inline hipError_t launch_gemm8r_act(const GemmArgs<T>& g, hipStream_t stream) {
    return g.epi.residual ? launch_gemm8r_inst<T, ACT, true>(g, stream) : launch_gemm8r_inst<T, ACT, false>(g, stream);
}

template <typename T>
inline hipError_t launch_gemm8r(const GemmArgs<T>& g, hipStream_t stream) {
    switch (g.epi.act) {
        case ACT_GELU_TANH: return launch_gemm8r_act<T, ACT_GELU_TANH>(g, stream);
        case ACT_GELU_ERF: return launch_gemm8r_act<T, ACT_GELU_ERF>(g, stream);
        default: return launch_gemm8r_act<T, ACT_NONE>(g, stream);
    }
}

}  // namespace zett
