// gemm.hip.h — MFMA GEMM for gfx950 (CDNA4) with a fused row-wise epilogue.
//
//   C[M,N] = epilogue( A[M,K] · W[N,K]ᵀ )
//
// Both operands are K-contiguous ("B-transposed" form: torch Linear weights are
// stored [out,in] so no transposition is needed).  One kernel body serves both
// arithmetic modes of the library:
//   T = bf16  -> v_mfma_f32_32x32x16_bf16  (8 bf16 of K per lane and instruction)
//   T = float -> v_mfma_f32_32x32x2_f32    (exact fp32; a 16-byte LDS chunk feeds 4
//                instructions, the K permutation is the same for A and W)
// The LDS image is identical in both modes: rows of 128 bytes of K, 16-byte chunks
// XOR-swizzled by (row>>1)&7 so that a ds_read_b128 lane group (16 lanes, 16
// distinct rows) touches 16 distinct 16-byte slots of the 256-byte bank row.
//
// Reduction order over K is fixed by the tile loop and never depends on M or on the
// position of a row inside the launch, so one input row always produces bit-identical
// output — the property the vocab-sharded multi-GPU path relies on.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// M tiles per group of the tile order of the large-tile kernels (gemm8r / gemm8x / gemm4d): consecutive workgroups of
// an XCD walk GROUP_M row tiles down before moving one column tile right, so the 32 tiles an XCD runs at a time form
// a GROUP_M x (32 / GROUP_M) block and share GROUP_M + 32 / GROUP_M operand panels in its L2.  4 x 8 and 8 x 4 are the
// same traffic, but the tall-skinny launches of this path (M = 30-80 k rows, N = 4-12 k) run 3-8 % faster on 4 x 8
// (tools/gemm_bench -DZETT_GROUP_M=2/4/8/16: 16 loses 6-9 %, 2 is within 2 % of 4 either way).
#ifndef ZETT_GROUP_M
#define ZETT_GROUP_M 4
#endif

namespace zett {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

typedef uint16_t bf16_t;   // storage type of bf16 activations / weights
typedef _Float16 f16_t;    // IEEE half: the third operand type (ZETT_PREC_F16)
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {   // round to nearest even: v_cvt_pk_bf16_f32 on gfx950
    return __builtin_bit_cast(bf16_t, (__bf16)f);
}

// fp32 -> operand type of a GEMM (round to nearest even)
template <typename T> __device__ __forceinline__ T to_lo(float v);
template <> __device__ __forceinline__ float to_lo<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t to_lo<bf16_t>(float v) { return f32_to_bf16(v); }
template <> __device__ __forceinline__ f16_t to_lo<f16_t>(float v) { return (f16_t)v; }
template <typename T> __device__ __forceinline__ float lo_to_f32(T v);
template <> __device__ __forceinline__ float lo_to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float lo_to_f32<bf16_t>(bf16_t v) { return bf16_to_f32(v); }
template <> __device__ __forceinline__ float lo_to_f32<f16_t>(f16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ uint32_t pack2_lo(float a, float b);      // two 16-bit operands in a dword
template <> __device__ __forceinline__ uint32_t pack2_lo<bf16_t>(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));      // one v_cvt_pk_bf16_f32
}
template <> __device__ __forceinline__ uint32_t pack2_lo<f16_t>(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    const h2 v = {(f16_t)a, (f16_t)b};
    return __builtin_bit_cast(uint32_t, v);
}
template <typename T> __device__ __forceinline__ void unpack2_lo(uint32_t u, float& a, float& b);
template <> __device__ __forceinline__ void unpack2_lo<bf16_t>(uint32_t u, float& a, float& b) { a = __uint_as_float(u << 16); b = __uint_as_float(u & 0xffff0000u); }
template <> __device__ __forceinline__ void unpack2_lo<f16_t>(uint32_t u, float& a, float& b) {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    const h2 v = __builtin_bit_cast(h2, u);
    a = (float)v[0]; b = (float)v[1];
}

enum { ACT_NONE = 0, ACT_GELU_TANH = 1, ACT_GELU_ERF = 2 };

// GELUs of the epilogues.  Written with v_exp_f32 / v_rcp_f32 based forms (absolute error
// ~1e-7, i.e. fp32 round-off class) instead of the libm tanhf / erff call sequences, which
// cost 4-5x as many VALU instructions in an epilogue that has to process 128 values per lane.
//
// Every epilogue formula is written with explicit fused multiply-adds under
// `#pragma clang fp contract(off)`: HIP's default -ffp-contract=fast lets the backend fuse a
// multiply and an add wherever it sees fit, and it chooses differently for the scalar
// epilogue of the 128x128 kernel and the float4 (packed-math) epilogues of the large tiles.
// The kernels must agree bit for bit (the tile is chosen per launch from M, so a vocabulary
// shard and the whole vocabulary can take different kernels for the same row).
__device__ __forceinline__ float gelu_tanh_f(float x) {   // F.gelu(approximate="tanh")
#pragma clang fp contract(off)
    const float x3 = (x * x) * x;
    const float u2 = 1.5957691216057308f * __builtin_fmaf(0.044715f, x3, x);        // 2 * sqrt(2/pi) * (x + 0.044715 x^3)
    // tanh(u) = 1 - 2 / (1 + exp(2u)); exp overflow -> tanh = 1, underflow -> -1
    const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u2 * 1.4426950408889634f));
    return (0.5f * x) * (1.0f + t);
}
__device__ __forceinline__ float erf_as_f(float x) {      // Abramowitz & Stegun 7.1.26, |err| <= 1.5e-7
#pragma clang fp contract(off)
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.0f));
    float p = 1.061405429f;
    p = __builtin_fmaf(p, t, -1.453152027f);
    p = __builtin_fmaf(p, t, 1.421413741f);
    p = __builtin_fmaf(p, t, -0.284496736f);
    p = __builtin_fmaf(p, t, 0.254829592f);
    const float ex = __builtin_amdgcn_exp2f((ax * ax) * -1.4426950408889634f);      // exp(-x^2)
    const float y = __builtin_fmaf(-(p * t), ex, 1.0f);
    return copysignf(y, x);
}
__device__ __forceinline__ float gelu_erf_f(float x) {    // F.gelu (erf form)
#pragma clang fp contract(off)
    return (x * 0.5f) * (1.0f + erf_as_f(x * 0.70710678118654752440f));
}

// LayerNorm's affine step, the ONE definition shared by the LayerNorm kernel (which writes the 16-bit operand of the
// next GEMM and two statistics per row) and by every consumer that needs the fp32 LayerNorm output again — the
// residual epilogues below and the position-0 readout: they recompute it from the pre-LayerNorm sum they read
// anyway, so the fp32 LayerNorm output is never stored (4 of the 10 bytes per element the LayerNorm kernel moved).
__device__ __forceinline__ float ln_affine(float x, float mean, float rstd, float g, float b) {
#pragma clang fp contract(off)
    return __builtin_fmaf((x - mean) * rstd, g, b);
}

// One output element of every GEMM epilogue (see GemmEpilogue below): the single definition all
// tile variants call, component by component.
template <int ACT>
__device__ __forceinline__ float epi_value(float acc, float bias, bool has_res, float res, bool has_scale, float sc, float sh) {
#pragma clang fp contract(off)
    float v = acc + bias;
    if (ACT == 1) v = gelu_tanh_f(v);
    else if (ACT == 2) v = gelu_erf_f(v);
    if (has_res) v = v + res;
    if (has_scale) v = __builtin_fmaf(sc, v, sh);
    return v;
}
template <int ACT>
__device__ __forceinline__ float4 epi_value4(float4 a, float4 bias, bool has_res, float4 res, bool has_scale, float4 sc, float4 sh) {
    return make_float4(epi_value<ACT>(a.x, bias.x, has_res, res.x, has_scale, sc.x, sh.x), epi_value<ACT>(a.y, bias.y, has_res, res.y, has_scale, sc.y, sh.y),
                       epi_value<ACT>(a.z, bias.z, has_res, res.z, has_scale, sc.z, sh.z), epi_value<ACT>(a.w, bias.w, has_res, res.w, has_scale, sc.w, sh.w));
}

// The same epilogue over NV values of one lane, written stage by stage on pairs of values (the packed fp32
// operations of the vector ALU): every statement below is one operation of epi_value applied to all NV values
// before the next operation starts, so the NV/2 dependency chains interleave.  The element-at-a-time form above
// compiles to one dependent chain of packed operations per pair of values, ~100 cycles per value with one wave per
// SIMD, and the machine scheduler rebuilds that order from any source order (it minimises live ranges; a
// sched_barrier does not hold pure arithmetic in place): the empty volatile asm statements between the stages do,
// because they keep their own order — a pair can run at most one stage ahead of the others.
// Operation for operation the arithmetic of epi_value: identical bits.
typedef __attribute__((ext_vector_type(2))) float f32x2;
#define ZETT_PIN_STAGE(arr, n) do { _Pragma("unroll") for (int e_ = 0; e_ < (n); ++e_) asm volatile("" : "+v"((arr)[e_])); } while (0)

template <int ACT, bool HAS_RES, bool HAS_SCALE, int NV>
__device__ __forceinline__ void epi_values(float (&vs)[NV], const float (&bias)[NV], const float (&res)[NV], const float (&sc)[NV], const float (&sh)[NV]) {
#pragma clang fp contract(off)
    static_assert(NV % 2 == 0, "pairs");
    constexpr int NP = NV / 2;
    f32x2 v[NP];
#pragma unroll
    for (int e = 0; e < NP; ++e) v[e] = f32x2{vs[2 * e], vs[2 * e + 1]} + f32x2{bias[2 * e], bias[2 * e + 1]};
    if (ACT == ACT_GELU_TANH) {
        f32x2 u2[NP], t[NP];
        ZETT_PIN_STAGE(v, NP);
#pragma unroll
        for (int e = 0; e < NP; ++e) u2[e] = (v[e] * v[e]) * v[e];
        ZETT_PIN_STAGE(u2, NP);
#pragma unroll
        for (int e = 0; e < NP; ++e) u2[e] = 1.5957691216057308f * __builtin_elementwise_fma(f32x2{0.044715f, 0.044715f}, u2[e], v[e]);
        ZETT_PIN_STAGE(u2, NP);
#pragma unroll
        for (int e = 0; e < NP; ++e) u2[e] = u2[e] * 1.4426950408889634f;
        ZETT_PIN_STAGE(u2, NP);
#pragma unroll
        for (int e = 0; e < NP; ++e) t[e] = f32x2{__builtin_amdgcn_exp2f(u2[e].x), __builtin_amdgcn_exp2f(u2[e].y)};
        ZETT_PIN_STAGE(t, NP);
#pragma unroll
        for (int e = 0; e < NP; ++e) t[e] = 1.0f + t[e];
        ZETT_PIN_STAGE(t, NP);
#pragma unroll
        for (int e = 0; e < NP; ++e) t[e] = f32x2{__builtin_amdgcn_rcpf(t[e].x), __builtin_amdgcn_rcpf(t[e].y)};
        ZETT_PIN_STAGE(t, NP);
#pragma unroll
        for (int e = 0; e < NP; ++e) t[e] = 1.0f - 2.0f * t[e];
        ZETT_PIN_STAGE(t, NP);
#pragma unroll
        for (int e = 0; e < NP; ++e) v[e] = (0.5f * v[e]) * (1.0f + t[e]);
        ZETT_PIN_STAGE(v, NP);
    } else if (ACT == ACT_GELU_ERF) {
        f32x2 z[NP], ax[NP], t[NP], p[NP], ex[NP];
        ZETT_PIN_STAGE(v, NP);
#pragma unroll
        for (int e = 0; e < NP; ++e) z[e] = v[e] * 0.70710678118654752440f;
        ZETT_PIN_STAGE(z, NP);
#pragma unroll
        for (int e = 0; e < NP; ++e) ax[e] = __builtin_elementwise_abs(z[e]);
#pragma unroll
        for (int e = 0; e < NP; ++e) t[e] = __builtin_elementwise_fma(f32x2{0.3275911f, 0.3275911f}, ax[e], f32x2{1.0f, 1.0f});
        ZETT_PIN_STAGE(t, NP);
#pragma unroll
        for (int e = 0; e < NP; ++e) t[e] = f32x2{__builtin_amdgcn_rcpf(t[e].x), __builtin_amdgcn_rcpf(t[e].y)};
        ZETT_PIN_STAGE(t, NP);
#pragma unroll
        for (int e = 0; e < NP; ++e) ex[e] = (ax[e] * ax[e]) * -1.4426950408889634f;
        ZETT_PIN_STAGE(ex, NP);
#pragma unroll
        for (int e = 0; e < NP; ++e) ex[e] = f32x2{__builtin_amdgcn_exp2f(ex[e].x), __builtin_amdgcn_exp2f(ex[e].y)};
        ZETT_PIN_STAGE(ex, NP);
#pragma unroll
        for (int e = 0; e < NP; ++e) p[e] = __builtin_elementwise_fma(f32x2{1.061405429f, 1.061405429f}, t[e], f32x2{-1.453152027f, -1.453152027f});
        ZETT_PIN_STAGE(p, NP);
#pragma unroll
        for (int e = 0; e < NP; ++e) p[e] = __builtin_elementwise_fma(p[e], t[e], f32x2{1.421413741f, 1.421413741f});
        ZETT_PIN_STAGE(p, NP);
#pragma unroll
        for (int e = 0; e < NP; ++e) p[e] = __builtin_elementwise_fma(p[e], t[e], f32x2{-0.284496736f, -0.284496736f});
        ZETT_PIN_STAGE(p, NP);
#pragma unroll
        for (int e = 0; e < NP; ++e) p[e] = __builtin_elementwise_fma(p[e], t[e], f32x2{0.254829592f, 0.254829592f});
        ZETT_PIN_STAGE(p, NP);
#pragma unroll
        for (int e = 0; e < NP; ++e) p[e] = -(p[e] * t[e]);
        ZETT_PIN_STAGE(p, NP);
#pragma unroll
        for (int e = 0; e < NP; ++e) p[e] = __builtin_elementwise_fma(p[e], ex[e], f32x2{1.0f, 1.0f});
        ZETT_PIN_STAGE(p, NP);
#pragma unroll
        for (int e = 0; e < NP; ++e) p[e] = 1.0f + __builtin_elementwise_copysign(p[e], z[e]);
        ZETT_PIN_STAGE(p, NP);
#pragma unroll
        for (int e = 0; e < NP; ++e) v[e] = (v[e] * 0.5f) * p[e];
        ZETT_PIN_STAGE(v, NP);
    }
    if (HAS_RES) {
#pragma unroll
        for (int e = 0; e < NP; ++e) v[e] = v[e] + f32x2{res[2 * e], res[2 * e + 1]};
    }
    if (HAS_SCALE) {
#pragma unroll
        for (int e = 0; e < NP; ++e) v[e] = __builtin_elementwise_fma(f32x2{sc[2 * e], sc[2 * e + 1]}, v[e], f32x2{sh[2 * e], sh[2 * e + 1]});
    }
#pragma unroll
    for (int e = 0; e < NP; ++e) { vs[2 * e] = v[e].x; vs[2 * e + 1] = v[e].y; }
}

// Row-wise epilogue description (all pointers device, nullable unless noted).
//   v = acc + bias[col]; v = act(v); v += r[row, col]; v = scale[col]*v + shift[col]
//   r = residual[row, col], or, with res_stats, the LayerNorm of that row recomputed on the fly:
//       r = ln_affine(residual[row, col], res_stats[2*row], res_stats[2*row + 1], res_gamma[col], res_beta[col])
//   (with res_index, `row` on the right-hand sides is res_index[row])
//   col <  split_col -> out_f32[row*ld_f32 + col], out_lo[row*ld_lo + col]
//   col >= split_col -> out_f32_b[row*ld_f32 + (col - split_col)]
template <typename T>
struct GemmEpilogue {
    const float* bias;
    int act;
    const float* residual;
    int ld_res;
    const float* res_stats;     // [M][2] (mean, rstd) of the residual rows, or null: the residual is used as stored
    const float* res_gamma;     // [N]
    const float* res_beta;      // [N]
    const int32_t* res_index;   // [M] row of `residual` / `res_stats` that output row m adds, or null: row m itself (layer 0
                                //     of the encoder: the residual stream starts per (source id, position) pair)
    // LayerNorm folded into the GEMMs on either side of it (gemm4d only; DESIGN.md §4 "LayerNorm fold"):
    float2* stats_part;         // producer: [N/128][ld_part] (sum, sum of squares) of the fp32 output row over each 128-column
    int ld_part;                //           slice; the launch also writes out_lo = the 16-bit copy of the same values
    const float* fold_stats;    // consumer: [M][2] (mean, rstd) of the A rows: the accumulator becomes rstd * (acc - mean * fold_c[n])
    const float* fold_c;        //           [N] row sums of the gamma-folded weight
    const float* scale;
    const float* shift;
    float* out_f32;
    int ld_f32;
    T* out_lo;
    int ld_lo;
    int split_col;
    float* out_f32_b;
    // Range guard (include/zett_hip.h zett_check_range): where a launch ORs its findings, or null.  16-bit outputs of an f16
    // launch are checked against the half range (ZETT_RANGE_ACTIVATION); with range_final the fp32 outputs are the
    // predicted embeddings and are checked for inf / NaN (ZETT_RANGE_OUTPUT).
    int32_t* range_flag;
    int range_final;
    // 16-bit residual stream (r4; gemm4d's LN16 producer only): the residual rows are read from the 16-bit copy of the hidden
    // state (type T) instead of `residual`; res_stats / res_gamma / res_beta / res_index apply to it as they do to `residual`
    const T* residual_lo;
    int ld_res_lo;
};

// Largest finite value of the operand type: what a value written as a 16-bit operand is checked against.  bf16 and fp32
// share fp32's exponent range: only the half type can overflow where the fp32 value was finite.
template <typename T> struct LoRange { static constexpr bool checked = false; static constexpr float limit = 0.f; };
template <> struct LoRange<f16_t> { static constexpr bool checked = true; static constexpr float limit = 65504.0f; };
// true where v would not survive: beyond the limit, or NaN (the comparison is false for NaN)
__device__ __forceinline__ bool out_of_range(float v, float limit) { return !(__builtin_fabsf(v) <= limit); }
constexpr float ZETT_F32_MAX = 3.402823466e38f;
constexpr int ZETT_RANGE_BIT_SOURCE = 1, ZETT_RANGE_BIT_ACTIVATION = 2, ZETT_RANGE_BIT_OUTPUT = 4, ZETT_RANGE_BIT_WEIGHT = 8;
__device__ __forceinline__ void range_report(int32_t* flag, bool bad, int bit) {
    if (bad && flag) atomicOr(flag, bit);      // (never taken on a healthy checkpoint)
}

template <typename T>
struct GemmArgs {
    const T* A;
    int lda;     // elements
    const T* W;
    int ldw;     // elements
    int M, N, K;
    GemmEpilogue<T> epi;
    int tile_order = 0;     // gemm4d only: 0 = column-tile-major groups (default), 1 = row-tile-major groups (A/B option)
    int group = 0;          // gemm4d only: column tiles per group (order 0) / row tiles per group (order 1); 0 = the default (4)
    int row0 = 0;           // gemm4d only.  To the launcher: -1 = cut a partly filled last round into 128x256 tiles (gemm4d_row_split), 0 = 256x256
                            // tiles only, > 0 = the caller's cut, -2 = 128x256 tiles only.  To the kernel: the first row its tiles cover
};

constexpr int GEMM_BM = 128;
constexpr int GEMM_BN = 128;
constexpr int GEMM_WAIT_VMCNT0 = 0x0F70;            // s_waitcnt vmcnt(0) expcnt(7) lgkmcnt(15) (gfx9 encoding)
constexpr int GEMM_ROW_BYTES = 128;                 // K bytes per tile row
constexpr int GEMM_TILE_BYTES = GEMM_BM * GEMM_ROW_BYTES;   // 16 KiB per operand tile

__device__ __forceinline__ int lds_chunk_off(int row, int chunk) {
    return row * GEMM_ROW_BYTES + ((chunk ^ ((row >> 1) & 7)) << 4);
}

template <typename T>
__device__ __forceinline__ void mfma_chunk(const u32x4& a, const u32x4& b, f32x16& acc);

template <>
__device__ __forceinline__ void mfma_chunk<bf16_t>(const u32x4& a, const u32x4& b, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                  acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mfma_chunk<f16_t>(const u32x4& a, const u32x4& b, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mfma_chunk<float>(const u32x4& a, const u32x4& b, f32x16& acc) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[j]), __uint_as_float(b[j]), acc, 0, 0, 0);
}

// 128x128 output tile, 256 threads = 4 waves in a 2x2 grid, each wave 64x64 =
// 2x2 MFMA tiles of 32x32.  Register-staged global->LDS with two LDS buffers: the
// loads of K-tile t+1 are in flight while K-tile t is multiplied.
template <typename T>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmArgs<T> g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int EPC = 16 / (int)sizeof(T);            // elements per 16-byte chunk
    constexpr int BK = GEMM_ROW_BYTES / (int)sizeof(T);  // K elements per tile

    // XCD-aware tile order: consecutive workgroups of one XCD share the W panel.
    const int tiles_m = (g.M + GEMM_BM - 1) / GEMM_BM;
    const int tiles_n = (g.N + GEMM_BN - 1) / GEMM_BN;
    const int nwg = tiles_m * tiles_n;
    int wg = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
    }
    // grouped order inside the XCD's range: 8 row-tiles x 8 col-tiles of concurrently
    // resident workgroups share 8 A panels + 8 W panels in that XCD's L2.
    constexpr int GROUP_M = 8;
    const int group_size = GROUP_M * tiles_n;
    const int first_m = (wg / group_size) * GROUP_M;
    const int gm = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
    const int tm = first_m + (wg % group_size) % gm;
    const int tn = (wg % group_size) / gm;
    const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;

    // global staging: 4 chunks of A and 4 of W per thread
    const unsigned char* a_src[4];
    const unsigned char* w_src[4];
    int lds_dst[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + 256 * i;
        const int row = c >> 3, ch = c & 7;
        int ar = m0 + row; ar = ar < g.M ? ar : g.M - 1;
        int wr = n0 + row; wr = wr < g.N ? wr : g.N - 1;
        a_src[i] = (const unsigned char*)(g.A + (size_t)ar * g.lda) + ch * 16;
        w_src[i] = (const unsigned char*)(g.W + (size_t)wr * g.ldw) + ch * 16;
        lds_dst[i] = lds_chunk_off(row, ch);
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = g.K / BK;
    u32x4 ra[4], rw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ra[i] = *(const u32x4*)(a_src[i]);
        rw[i] = *(const u32x4*)(w_src[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        *(u32x4*)(smem + lds_dst[i]) = ra[i];
        *(u32x4*)(smem + GEMM_TILE_BYTES + lds_dst[i]) = rw[i];
    }
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const unsigned char* As = smem + cur * 2 * GEMM_TILE_BYTES;
        const unsigned char* Ws = As + GEMM_TILE_BYTES;
        const bool more = kt + 1 < nk;
        if (more) {
            const size_t koff = (size_t)(kt + 1) * GEMM_ROW_BYTES;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i] = *(const u32x4*)(a_src[i] + koff);
                rw[i] = *(const u32x4*)(w_src[i] + koff);
            }
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int ch = kk * 2 + hi;
            u32x4 fa[2], fw[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = *(const u32x4*)(As + lds_chunk_off(wm * 64 + i * 32 + l31, ch));
                fw[i] = *(const u32x4*)(Ws + lds_chunk_off(wn * 64 + i * 32 + l31, ch));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mfma_chunk<T>(fa[i], fw[j], acc[i][j]);
        }
        if (more) {
            unsigned char* Ad = smem + (cur ^ 1) * 2 * GEMM_TILE_BYTES;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *(u32x4*)(Ad + lds_dst[i]) = ra[i];
                *(u32x4*)(Ad + GEMM_TILE_BYTES + lds_dst[i]) = rw[i];
            }
        }
        __syncthreads();
    }
    (void)EPC;

    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31,
    //      row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const GemmEpilogue<T>& e = g.epi;
    bool bad_lo = false, bad_out = false;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + l31;
        if (col >= g.N) continue;
        const float bias = e.bias ? e.bias[col] : 0.f;
        const float sc = e.scale ? e.scale[col] : 1.f;
        const float sh = e.shift ? e.shift[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (row >= g.M) continue;
                const size_t rrow = e.res_index ? (size_t)e.res_index[row] : (size_t)row;
                float res = e.residual ? e.residual[rrow * e.ld_res + col] : 0.f;
                if (e.residual && e.res_stats) res = ln_affine(res, e.res_stats[2 * rrow], e.res_stats[2 * rrow + 1], e.res_gamma[col], e.res_beta[col]);
                const bool hr = e.residual != nullptr, hs = e.scale != nullptr;
                const float v = e.act == ACT_GELU_TANH ? epi_value<ACT_GELU_TANH>(acc[i][j][r], bias, hr, res, hs, sc, sh)
                              : e.act == ACT_GELU_ERF ? epi_value<ACT_GELU_ERF>(acc[i][j][r], bias, hr, res, hs, sc, sh)
                                                      : epi_value<ACT_NONE>(acc[i][j][r], bias, hr, res, hs, sc, sh);
                if (col < e.split_col) {
                    if (e.out_f32) e.out_f32[(size_t)row * e.ld_f32 + col] = v;
                    if (e.out_lo) {
                        e.out_lo[(size_t)row * e.ld_lo + col] = to_lo<T>(v);
                        if (LoRange<T>::checked) bad_lo |= out_of_range(v, LoRange<T>::limit);
                    }
                } else if (e.out_f32_b) {
                    e.out_f32_b[(size_t)row * e.ld_f32 + (col - e.split_col)] = v;
                }
                if (e.range_final) bad_out |= out_of_range(v, ZETT_F32_MAX);
            }
        }
    }
    range_report(e.range_flag, bad_lo, ZETT_RANGE_BIT_ACTIVATION);
    range_report(e.range_flag, bad_out, ZETT_RANGE_BIT_OUTPUT);
}

constexpr int GEMM_LDS_BYTES = 4 * GEMM_TILE_BYTES;   // two buffers x (A tile + W tile) = 64 KiB

template <typename T>
inline hipError_t launch_gemm(const GemmArgs<T>& g, hipStream_t stream) {
    const int tiles_m = (g.M + GEMM_BM - 1) / GEMM_BM;
    const int tiles_n = (g.N + GEMM_BN - 1) / GEMM_BN;
    if (tiles_m <= 0 || tiles_n <= 0) return hipSuccess;
    hipLaunchKernelGGL(gemm_tn_kernel<T>, dim3(tiles_m * tiles_n), dim3(256), GEMM_LDS_BYTES, stream, g);
    return hipGetLastError();
}

}  // namespace zett
