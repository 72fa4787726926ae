// gemm256r.hip.h — 256x256 MFMA GEMM with a deep LDS-DMA ring (gfx950).
//
//   C[M,N] = epilogue( A[M,K] · W[N,K]ᵀ )      (contract and epilogue of gemm.hip.h)
//
// Why a ring: with two 64 KiB stages only one DMA batch (64 KiB per CU) is ever in
// flight, and the measured L2->LDS rate with one batch in flight (~11.5 TB/s chip-wide,
// tools/gemm_bench ablations) caps the kernel near 1.1 PFLOP/s whatever the MFMA schedule
// does.  Here the K loop advances in half steps of 64 bytes per row; a half step of both
// operands is a 32 KiB slot and NS slots (4 = 128 KiB or 5 = 160 KiB) form a ring, so
// NS-1 slots (96 / 128 KiB per CU) are in flight under the MFMAs and a slot is requested
// NS-1 half steps before it is read.  Waits are counted (s_waitcnt vmcnt(4*(NS-2))), never
// a drain, and there is one raw s_barrier per half step:
//
//   iteration h:  vmcnt(4*(NS-2))  -> my share of half step h+1 has landed
//                 lgkmcnt(0), s_barrier -> everybody's share has; slot h%NS is dead
//                 DMA(h+NS) -> slot h%NS
//                 12 ds_read_b128 of half step h+1 into the other fragment set
//                 16 MFMAs of half step h from registers
//
// LDS slot image: A rows then W rows, 64 bytes per row, 16-byte chunks XOR-swizzled by
// (row>>2)&3 (conflict-free for ds_read_b128: a 16-lane group touches 16 distinct slots of
// the 256-byte bank row); the DMA applies the same XOR to the source chunk of each lane.
// EXPERIMENT (not part of libzett_hip.so): built only by tools/gemm_bench.hip; measured
// results and why the product does not use it are in DESIGN.md §4.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm256.hip.h"

namespace zett {

constexpr int R_ROW_BYTES = 64;
constexpr int R_OPERAND_BYTES = 256 * R_ROW_BYTES;     // 16 KiB
constexpr int R_SLOT_BYTES = 2 * R_OPERAND_BYTES;      // 32 KiB

template <int N> __device__ __forceinline__ void wait_vmcnt_barrier() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// `newer` = number of ring slots requested after the one that must have landed
__device__ __forceinline__ void wait_landed(int newer) {
    if (newer >= 4) wait_vmcnt_barrier<16>();
    else if (newer == 3) wait_vmcnt_barrier<12>();
    else if (newer == 2) wait_vmcnt_barrier<8>();
    else if (newer == 1) wait_vmcnt_barrier<4>();
    else wait_vmcnt_barrier<0>();
}

template <typename T, int NS>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm256r_tn_kernel(GemmArgs<T> g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BKH = R_ROW_BYTES / (int)sizeof(T);       // K elements per half step

    const int tiles_m = (g.M + G256_BM - 1) / G256_BM;
    const int tiles_n = (g.N + G256_BN - 1) / G256_BN;
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
    const int m0 = tm * G256_BM, n0 = tn * G256_BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5;

    // DMA plan: a wave instruction covers 16 rows x 64 B; 2 instructions per operand and wave
    const unsigned char* a_src[2];
    const unsigned char* w_src[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = wave * 32 + j * 16 + (lane >> 2);
        const int ch = (lane & 3) ^ ((row >> 2) & 3);
        int ar = m0 + row; ar = ar < g.M ? ar : g.M - 1;
        int wr = n0 + row; wr = wr < g.N ? wr : g.N - 1;
        a_src[j] = (const unsigned char*)(g.A + (size_t)ar * g.lda) + ch * 16;
        w_src[j] = (const unsigned char*)(g.W + (size_t)wr * g.ldw) + ch * 16;
    }
    const int dma_base = wave * 32 * R_ROW_BYTES;

    auto issue_slot = [&](int h) {
        unsigned char* sa = smem + (h % NS) * R_SLOT_BYTES + dma_base;
        unsigned char* sw = sa + R_OPERAND_BYTES;
        const size_t koff = (size_t)h * R_ROW_BYTES;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a_src[j] + koff), (lds_ptr_t)(sa + j * 16 * R_ROW_BYTES), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(w_src[j] + koff), (lds_ptr_t)(sw + j * 16 * R_ROW_BYTES), 16, 0, 0);
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int a_off[4], w_off[2], swz[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) a_off[i] = (wm * 128 + i * 32 + l31) * R_ROW_BYTES;
#pragma unroll
    for (int j = 0; j < 2; ++j) w_off[j] = R_OPERAND_BYTES + (wn * 64 + j * 32 + l31) * R_ROW_BYTES;
    const int row_swz = (l31 >> 2) & 3;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) swz[kk] = ((kk * 2 + hi) ^ row_swz) << 4;

    u32x4 fa[2][2][4], fw[2][2][2];     // [set][kk][i|j]

    auto read_frags = [&](int h, int set) {
        const unsigned char* S = smem + (h % NS) * R_SLOT_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int j = 0; j < 2; ++j) fw[set][kk][j] = *(const u32x4*)(S + w_off[j] + swz[kk]);
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[set][kk][i] = *(const u32x4*)(S + a_off[i] + swz[kk]);
        }
    };
    auto multiply = [&](int set) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mfma_chunk<T>(fa[set][kk][i], fw[set][kk][j], acc[i][j]);
    };

    const int nh = g.K / BKH;
    const int pro = nh < NS ? nh : NS;
    for (int h = 0; h < pro; ++h) issue_slot(h);
    wait_landed(pro - 1);                      // half step 0 is in LDS
    read_frags(0, 0);

    auto iteration = [&](int h, int cur) {
        // newer requests than h+1 that are outstanding: h+2 .. min(h+NS-1, nh-1)
        int newer = nh - 2 - h;
        newer = newer < 0 ? 0 : (newer > NS - 2 ? NS - 2 : newer);
        wait_landed(newer);
        if (h + NS < nh) issue_slot(h + NS);
        if (h + 1 < nh) read_frags(h + 1, cur ^ 1);
        multiply(cur);
    };
    int h = 0;
    for (; h + 1 < nh; h += 2) {
        iteration(h, 0);
        iteration(h + 1, 1);
    }
    if (h < nh) iteration(h, 0);

    // ---- epilogue (see gemm256.hip.h)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    float* region = (float*)(smem + wave * 16384);
    const GemmEpilogue<T>& e = g.epi;
    const int c4 = (lane & 15) * 4;
    const int gcol = n0 + wn * 64 + c4;
    const bool col_ok = gcol < g.N;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f), sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = bias4;
    if (col_ok) {
        if (e.bias) bias4 = *(const float4*)(e.bias + gcol);
        if (e.scale) sc4 = *(const float4*)(e.scale + gcol);
        if (e.shift) sh4 = *(const float4*)(e.shift + gcol);
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    region[(i2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * 64 + j * 32 + l31] = acc[2 * p + i2][j][r];
        for (int t = 0; t < 16; ++t) {
            const int lrow = t * 4 + (lane >> 4);
            const int grow = m0 + wm * 128 + p * 64 + lrow;
            float4 v = *(const float4*)(region + lrow * 64 + c4);
            if (grow >= g.M || !col_ok) continue;
            v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
            if (e.act == ACT_GELU_TANH) { v.x = gelu_tanh_f(v.x); v.y = gelu_tanh_f(v.y); v.z = gelu_tanh_f(v.z); v.w = gelu_tanh_f(v.w); }
            else if (e.act == ACT_GELU_ERF) { v.x = gelu_erf_f(v.x); v.y = gelu_erf_f(v.y); v.z = gelu_erf_f(v.z); v.w = gelu_erf_f(v.w); }
            if (e.residual) {
                const float4 rr = *(const float4*)(e.residual + (size_t)grow * e.ld_res + gcol);
                v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
            }
            if (e.scale) { v.x = sc4.x * v.x + sh4.x; v.y = sc4.y * v.y + sh4.y; v.z = sc4.z * v.z + sh4.z; v.w = sc4.w * v.w + sh4.w; }
            if (gcol < e.split_col) {
                if (e.out_f32) *(float4*)(e.out_f32 + (size_t)grow * e.ld_f32 + gcol) = v;
                if (e.out_lo) store_out4<T>(e.out_lo + (size_t)grow * e.ld_lo + gcol, v);
            } else if (e.out_f32_b) {
                *(float4*)(e.out_f32_b + (size_t)grow * e.ld_f32 + (gcol - e.split_col)) = v;
            }
        }
    }
}

template <typename T, int NS>
inline hipError_t launch_gemm256r(const GemmArgs<T>& g, hipStream_t stream) {
    const int tiles_m = (g.M + G256_BM - 1) / G256_BM;
    const int tiles_n = (g.N + G256_BN - 1) / G256_BN;
    if (tiles_m <= 0 || tiles_n <= 0) return hipSuccess;
    hipLaunchKernelGGL((gemm256r_tn_kernel<T, NS>), dim3(tiles_m * tiles_n), dim3(512), NS * R_SLOT_BYTES, stream, g);
    return hipGetLastError();
}

}  // namespace zett
