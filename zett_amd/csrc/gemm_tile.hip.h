// gemm_tile.hip.h — what the large-tile GEMM kernels (gemm8r, gemm384, gemm4d) share: the geometry of the 256x256
// tile and its LDS image, the 16-bit / fp32 output stores and the epilogue drain (EpiDrain).
//
// LDS image per operand and stage: 256 rows x 128 B of K, 16-byte chunks XOR-swizzled by (row>>1)&7 (as in
// gemm.hip.h).  LDS-DMA writes are lane-linear (wave-uniform base + lane*16), so where a kernel fills the image by
// LDS-DMA the swizzle is applied to each lane's SOURCE address (the 8 lanes of a row permute the 8 chunks of one
// 128-byte line: still one coalesced line per row) and undone by the same XOR on the ds_read_b128 side.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm.hip.h"

namespace zett {

// Per-device opt-in to > 64 KiB of dynamic LDS: the attribute belongs to the function ON A DEVICE, and one process may
// drive several (one handle per device).
struct DeviceFlags {
    bool set[64] = {};
    bool* current() {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return nullptr;
        return &set[d];
    }
};

constexpr int G4R_RSRC_WORD3 = 0x00020000;   // raw buffer, DATA_FORMAT_32 (gfx9 resource word 3)
constexpr int G4R_WAIT_LGKM0 = 0xC07F;       // s_waitcnt lgkmcnt(0), vmcnt/expcnt untouched

constexpr int G256_BM = 256;
constexpr int G256_BN = 256;
constexpr int G256_OPERAND_BYTES = G256_BM * GEMM_ROW_BYTES;      // 32 KiB
constexpr int G256_STAGE_BYTES = 2 * G256_OPERAND_BYTES;          // A + W = 64 KiB
constexpr int G256_LDS_BYTES = 2 * G256_STAGE_BYTES;              // 128 KiB

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <typename T> __device__ __forceinline__ void store_out4(T* dst, float4 v);
template <> __device__ __forceinline__ void store_out4<float>(float* dst, float4 v) { *(float4*)dst = v; }
template <> __device__ __forceinline__ void store_out4<bf16_t>(bf16_t* dst, float4 v) {
    *(uint2*)dst = make_uint2(pack2_lo<bf16_t>(v.x, v.y), pack2_lo<bf16_t>(v.z, v.w));
}
template <> __device__ __forceinline__ void store_out4<f16_t>(f16_t* dst, float4 v) {
    *(uint2*)dst = make_uint2(pack2_lo<f16_t>(v.x, v.y), pack2_lo<f16_t>(v.z, v.w));
}

template <typename T> __device__ __forceinline__ void store_out8(T* dst, float4 a, float4 b);
template <> __device__ __forceinline__ void store_out8<float>(float* dst, float4 a, float4 b) { *(float4*)dst = a; *(float4*)(dst + 4) = b; }
template <> __device__ __forceinline__ void store_out8<bf16_t>(bf16_t* dst, float4 a, float4 b) {
    *(uint4*)dst = make_uint4(pack2_lo<bf16_t>(a.x, a.y), pack2_lo<bf16_t>(a.z, a.w), pack2_lo<bf16_t>(b.x, b.y), pack2_lo<bf16_t>(b.z, b.w));
}
template <> __device__ __forceinline__ void store_out8<f16_t>(f16_t* dst, float4 a, float4 b) {
    *(uint4*)dst = make_uint4(pack2_lo<f16_t>(a.x, a.y), pack2_lo<f16_t>(a.z, a.w), pack2_lo<f16_t>(b.x, b.y), pack2_lo<f16_t>(b.z, b.w));
}

// Epilogue drain shared by the large-tile kernels.  One wave has staged ROWS x COLS fp32
// accumulators (row stride COLS floats) in its private LDS region; every lane takes EIGHT
// consecutive columns of a row, so the 16-bit output is one global_store_dwordx4 per lane and
// instruction: stores are issue-bound (a wave instruction costs about the same whatever its
// width; MI355X_MICROARCH.md "store tail"), and with four columns per lane the bf16 output of a
// 256x256 tile took 10 us.  The two 16-byte LDS reads of a lane are ordered so that the lane
// groups of ds_read_b128 hit 16 distinct bank quads (rows are a multiple of 256 bytes apart):
// the second half first for odd rows (COLS = 64) / for the upper half of the row (COLS = 128).
//
// Order of a pass: residual rows requested first, all at once (load_res), the caller stages the
// accumulators, one explicit vmcnt(0), then a drain without any load in it.  Loads and stores
// share the vmcnt counter: a wait for a residual value placed between stores also waits for
// every earlier store to be acknowledged — one full write latency per row group, which the
// first version of this epilogue paid (11 us per tile).
// SCALE = false: the caller guarantees epi.scale == nullptr (saves the 16 scale/shift registers).
// NTF32 = false: never use the non-temporal store path (callers whose register budget is exhausted).
// GUARD = false: no range guard in this drain (gemm384: 168 registers per wave, the comparisons spill; its launches are
// covered by the catch-all check of the launches that write the predicted embeddings, which never take that tile).
template <typename T, int ACT, bool RES, int ROWS, int COLS, bool SCALE = true, bool NTF32 = true, bool GUARD = true>
struct EpiDrain {
    static constexpr int LPR = COLS / 8;       // lanes per row
    static constexpr int RPI = 64 / LPR;       // rows per wave instruction
    static constexpr int NIT = ROWS / RPI;     // instructions per pass

    static __device__ __forceinline__ void load_res(const GemmArgs<T>& g, int row0, int gcol, bool col_ok, int lane, float4 (&oa)[NIT], float4 (&ob)[NIT]) {
        if (!RES) return;
#pragma unroll
        for (int t = 0; t < NIT; ++t) {
            const int grow = row0 + t * RPI + lane / LPR;
            const bool ok = grow < g.M && col_ok;
            const size_t rrow = (g.epi.res_index && grow < g.M) ? (size_t)g.epi.res_index[grow] : (size_t)grow;
            const float* src = g.epi.residual + rrow * g.epi.ld_res + gcol;
            oa[t] = ok ? *(const float4*)src : make_float4(0.f, 0.f, 0.f, 0.f);
            ob[t] = ok ? *(const float4*)(src + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }

    // LayerNorm'd residual (GemmEpilogue::res_stats): the rows' statistics are requested with the residual rows,
    // gamma / beta of the lane's eight columns before the first pass (load_ln_cols: a load issued behind a store
    // could not be waited for without waiting for the store); ln_res applies the affine step to the staged rows
    // once everything has landed
    // (statistics: lane l of the wave holds those of row l of the pass — ROWS <= 64 — and the drain fetches the
    //  row of each instruction with a wave shuffle: two registers instead of two per instruction)
    static __device__ __forceinline__ void load_res_stats(const GemmArgs<T>& g, int row0, int lane, float2& st) {
        st = make_float2(0.f, 1.f);
        if (!RES || !g.epi.res_stats) return;
        static_assert(ROWS <= 64, "one statistics row per lane");
        int grow = row0 + (lane < ROWS ? lane : ROWS - 1);
        grow = grow < g.M ? grow : g.M - 1;
        if (g.epi.res_index) grow = g.epi.res_index[grow];
        st = *(const float2*)(g.epi.res_stats + 2 * (size_t)grow);
    }
    static __device__ __forceinline__ void load_ln_cols(const GemmEpilogue<T>& e, int gcol, bool col_ok, float4 (&gm)[2], float4 (&bt)[2]) {
        gm[0] = gm[1] = make_float4(1.f, 1.f, 1.f, 1.f);
        bt[0] = bt[1] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!RES || !e.res_stats || !col_ok) return;
        gm[0] = *(const float4*)(e.res_gamma + gcol); gm[1] = *(const float4*)(e.res_gamma + gcol + 4);
        bt[0] = *(const float4*)(e.res_beta + gcol); bt[1] = *(const float4*)(e.res_beta + gcol + 4);
    }
    static __device__ __forceinline__ void ln_res(const GemmArgs<T>& g, int lane, const float4 (&gm)[2], const float4 (&bt)[2], float2 st, float4 (&oa)[NIT], float4 (&ob)[NIT]) {
        if (!RES || !g.epi.res_stats) return;
        const float4 g0 = gm[0], g1 = gm[1], b0 = bt[0], b1 = bt[1];
#pragma unroll
        for (int t = 0; t < NIT; ++t) {
            const int lrow = t * RPI + lane / LPR;
            const float m = __shfl(st.x, lrow, 64), r = __shfl(st.y, lrow, 64);
            oa[t] = make_float4(ln_affine(oa[t].x, m, r, g0.x, b0.x), ln_affine(oa[t].y, m, r, g0.y, b0.y), ln_affine(oa[t].z, m, r, g0.z, b0.z), ln_affine(oa[t].w, m, r, g0.w, b0.w));
            ob[t] = make_float4(ln_affine(ob[t].x, m, r, g1.x, b1.x), ln_affine(ob[t].y, m, r, g1.y, b1.y), ln_affine(ob[t].z, m, r, g1.z, b1.z), ln_affine(ob[t].w, m, r, g1.w, b1.w));
        }
    }

    static __device__ __forceinline__ void drain(const GemmArgs<T>& g, const float* region, int row0, int gcol, bool col_ok, int lane,
                                                 const float4 (&bias8)[2], const float4 (&sc8)[2], const float4 (&sh8)[2], float4 (&oa)[NIT], float4 (&ob)[NIT]) {
        const GemmEpilogue<T>& e = g.epi;
        const int idx = lane % LPR;
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool has_scale = SCALE && e.scale != nullptr;
        // non-temporal fp32 stores for the residual epilogue of long-K launches (+3..6 % there: tools/gemm_bench
        // EPI=2; at K = 1024, where a CU is in an epilogue a third of the time, they cost 12 %)
        const bool nt_f32 = NTF32 && RES && g.K >= 2048;
#pragma unroll
        for (int t = 0; t < NIT; ++t) {
            const int lrow = t * RPI + lane / LPR;
            const int sw = COLS == 128 ? (idx >> 3) : (lrow & 1);
            const float* src = region + lrow * COLS + idx * 8;
            const float4 first = *(const float4*)(src + 4 * sw), second = *(const float4*)(src + 4 * (sw ^ 1));
            const float4 lo = sw ? second : first, hi = sw ? first : second;
            oa[t] = epi_value4<ACT>(lo, bias8[0], RES, RES ? oa[t] : zero, has_scale, sc8[0], sh8[0]);
            ob[t] = epi_value4<ACT>(hi, bias8[1], RES, RES ? ob[t] : zero, has_scale, sc8[1], sh8[1]);
        }
        // range guard (GemmEpilogue::range_flag): the values about to be written as 16-bit operands / as predicted embeddings
        if (GUARD && e.range_flag) {
            bool bad = false;
            const float lim = e.range_final ? ZETT_F32_MAX : LoRange<T>::limit;
            if (e.range_final || (LoRange<T>::checked && e.out_lo)) {
#pragma unroll
                for (int t = 0; t < NIT; ++t) {
                    const int grow = row0 + t * RPI + lane / LPR;
                    if (grow >= g.M || !col_ok) continue;
                    bad |= out_of_range(oa[t].x, lim) | out_of_range(oa[t].y, lim) | out_of_range(oa[t].z, lim) | out_of_range(oa[t].w, lim) |
                           out_of_range(ob[t].x, lim) | out_of_range(ob[t].y, lim) | out_of_range(ob[t].z, lim) | out_of_range(ob[t].w, lim);
                }
            }
            range_report(e.range_flag, bad, e.range_final ? ZETT_RANGE_BIT_OUTPUT : ZETT_RANGE_BIT_ACTIVATION);
        }
#pragma unroll
        for (int t = 0; t < NIT; ++t) {
            const int grow = row0 + t * RPI + lane / LPR;
            if (grow >= g.M || !col_ok) continue;
            if (gcol < e.split_col) {
                if (e.out_f32) {
                    float* d = e.out_f32 + (size_t)grow * e.ld_f32 + gcol;
                    if (nt_f32) {      // streamed once to the LayerNorm kernel: keep it out of the L2 the operand panels live in
                        // (inline asm: two IR stores that differ only in the nontemporal hint get merged into a plain one)
                        const f32x4 va = {oa[t].x, oa[t].y, oa[t].z, oa[t].w}, vb = {ob[t].x, ob[t].y, ob[t].z, ob[t].w};
                        // (s_nop 1: the two wait states hipcc itself leaves between a >8-byte store and a VALU write
                        //  of its data registers; it does not know this block is a store)
                        asm volatile("global_store_dwordx4 %0, %1, off nt\n\tglobal_store_dwordx4 %0, %2, off offset:16 nt\n\ts_nop 1"
                                     :: "v"(d), "v"(va), "v"(vb) : "memory");
                    } else {
                        *(float4*)d = oa[t];
                        *(float4*)(d + 4) = ob[t];
                    }
                }
                if (e.out_lo) store_out8<T>(e.out_lo + (size_t)grow * e.ld_lo + gcol, oa[t], ob[t]);
            } else if (e.out_f32_b) {
                float* d = e.out_f32_b + (size_t)grow * e.ld_f32 + (gcol - e.split_col);
                *(float4*)d = oa[t]; *(float4*)(d + 4) = ob[t];
            }
        }
    }

    // bias / scale / shift of the lane's eight columns
    static __device__ __forceinline__ void load_cols(const GemmEpilogue<T>& e, int gcol, bool col_ok, float4 (&bias8)[2], float4 (&sc8)[2], float4 (&sh8)[2]) {
        bias8[0] = bias8[1] = sh8[0] = sh8[1] = make_float4(0.f, 0.f, 0.f, 0.f);
        sc8[0] = sc8[1] = make_float4(1.f, 1.f, 1.f, 1.f);
        if (!col_ok) return;
        if (e.bias) { bias8[0] = *(const float4*)(e.bias + gcol); bias8[1] = *(const float4*)(e.bias + gcol + 4); }
        if (SCALE && e.scale) { sc8[0] = *(const float4*)(e.scale + gcol); sc8[1] = *(const float4*)(e.scale + gcol + 4); }
        if (SCALE && e.shift) { sh8[0] = *(const float4*)(e.shift + gcol); sh8[1] = *(const float4*)(e.shift + gcol + 4); }
    }
};

}  // namespace zett
