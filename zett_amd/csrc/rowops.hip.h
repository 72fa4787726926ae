// rowops.hip.h — the HBM-bound kernels of the hypernet forward (gfx950).
//
//   plan_*          surface-form matrix -> packed-token plan (pad skipping, distinct ids)
//   gather_src      A2/A3: source-embedding gather + in_scaler + fallback select
//   layernorm_rows  LayerNorm over H (+ the RobertaEmbeddings add for the embed variant)
//   attention_rows  per-row bidirectional attention over <= L' packed positions
//   (the position-0 readout + bias head is the last LayerNorm launch: layernorm_rows with LnReadout)
//
// All of them are one-pass streaming kernels: 16-byte vector accesses, one
// workgroup (or wave) per row, no inter-workgroup communication.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "gemm.hip.h"

namespace zett {

// ---------------------------------------------------------------------------
// Plan.  Reference semantics being preserved (modeling_hypernet.py:190, 220-234 and
// the eager RobertaSelfAttention mask):
//   * key j of row n is visible iff ids[n,j] != pad (the language token always is);
//   * only hidden[:,0] is read out, so a position matters only as a visible key or as
//     position 0 (a query even when it is pad);
//   * a row whose keys are ALL masked attends uniformly to every position (additive
//     finfo.min on every key), so such a row keeps all L positions, flagged uniform.
// Everything else (pad positions of ordinary rows) never influences an output and is
// not computed.
// ---------------------------------------------------------------------------

struct PlanArrays {
    int32_t* row_count;     // [N]   packed positions of row n
    int32_t* row_offset;    // [N+1] exclusive scan of row_count
    uint8_t* row_uniform;   // [N]
    int32_t* id_flag;       // [V]   1 if the id is referenced by a kept position
    int32_t* id_slot;       // [V+1] exclusive scan of id_flag; id_slot[V] = distinct ids
    int32_t* id_list;       // [<=V] slot -> id
    int32_t* tok_slot;      // [T]   table slot of the token, -1 = language token
    int32_t* tok_pos;       // [T]   position index (for position_embeddings)
    int32_t* tok_row;       // [T]   row the packed position belongs to
    // distinct (source id, position) pairs: what the embeddings' output — and so layer 0's Q/K/V — depends on
    int32_t* tok_pkey;      // [T]   id * Lp + position (Lp = L + lang; the language token counts as id V)
    int32_t* pair_flag;     // [(V+1)*Lp]   1 if the pair occurs (null: the lever is off for this call)
    int32_t* pair_slot;     // [(V+1)*Lp+1] exclusive scan of pair_flag; last = distinct pairs
    int32_t* tok_pair;      // [T]   pair slot of the packed position
    int32_t* pair_tslot;    // [P]   table slot of the pair (-1 = language token)
    int32_t* pair_pos;      // [P]   its position index
    uint8_t* tok_key;       // [T]   visible as key
    int32_t* err;           // [1]   set to 1 + row on an out-of-range id
    int32_t* counters;      // [3]   behind row_offset[N]: distinct ids, the error word, distinct pairs — what the host reads with the row offsets, in ONE copy
    int32_t n_pair_keys;    //       size of pair_flag (0: no pair plan)
};

__global__ void plan_rows_kernel(const int32_t* __restrict__ sfm, int64_t n_rows, int seq, int pad, int lam,
                                 int n_ids, PlanArrays p) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_rows) return;
    const int32_t* row = sfm + n * seq;
    int visible = 0;
    for (int j = 0; j < seq; ++j) {
        const int id = row[j];
        if (id < 0 || id >= n_ids) {          // F.embedding would raise IndexError here
            atomicMax(p.err, (int)(n < 0x7ffffffe ? n + 1 : 0x7fffffff));
            p.row_count[n] = 0;
            p.row_uniform[n] = 2;             // poisoned: skipped by plan_tokens_kernel
            return;
        }
        visible += (id != pad);
    }
    const bool uniform = (visible + lam) == 0;
    int count;
    if (uniform) {
        count = seq;
        for (int j = 0; j < seq; ++j) p.id_flag[row[j]] = 1;
    } else {
        count = visible + lam + (row[0] == pad ? 1 : 0);
        for (int j = 0; j < seq; ++j)
            if (j == 0 || row[j] != pad) p.id_flag[row[j]] = 1;
    }
    p.row_count[n] = count;
    p.row_uniform[n] = uniform ? 1 : 0;
}

// Single-workgroup exclusive scan (n <= a few million): out[i] = sum(in[0..i)), out[n] = total.
__global__ __launch_bounds__(1024) void exclusive_scan_kernel(const int32_t* __restrict__ in,
                                                              int32_t* __restrict__ out, int64_t n) {
    __shared__ int32_t part[1024];
    const int t = threadIdx.x;
    const int64_t per = (n + 1023) / 1024;
    const int64_t b = (int64_t)t * per, e = (b + per < n) ? b + per : n;
    int32_t s = 0;
    for (int64_t i = b; i < e; ++i) s += in[i];
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int32_t v = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int32_t run = (t == 0) ? 0 : part[t - 1];
    for (int64_t i = b; i < e; ++i) { const int32_t v = in[i]; out[i] = run; run += v; }
    if (t == 1023) out[n] = part[1023];
}

// Multi-block exclusive scan for the plan arrays (V can be 250 k source ids, N 262 k rows: the single
// workgroup above takes 75-400 us on those).  Chunks of SCAN_CHUNK elements:
//   (1) scan_chunk_sums_kernel: sums[c] = sum of chunk c;
//   (2) exclusive_scan_kernel on sums (a few hundred entries) -> offs[c], offs[chunks] = total;
//   (3) scan_apply_kernel: out[i] = offs[c] + exclusive scan inside the chunk; out[n] = total.
constexpr int SCAN_CHUNK = 4096;          // 256 threads x 16 elements

__device__ __forceinline__ int32_t block_exclusive_scan_256(int32_t v, int32_t* sh /* [256] */, int32_t* total) {
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int32_t add = (t >= off) ? sh[t - off] : 0;
        __syncthreads();
        sh[t] += add;
        __syncthreads();
    }
    if (total) *total = sh[255];
    return sh[t] - v;
}

__global__ __launch_bounds__(256) void scan_chunk_sums_kernel(const int32_t* __restrict__ in, int64_t n, int32_t* __restrict__ sums) {
    __shared__ int32_t sh[256];
    const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + threadIdx.x * 16;
    int32_t s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { const int64_t i = base + k; if (i < n) s += in[i]; }
    int32_t total;
    block_exclusive_scan_256(s, sh, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void scan_apply_kernel(const int32_t* __restrict__ in, int64_t n, const int32_t* __restrict__ offs,
                                                         int32_t* __restrict__ out) {
    __shared__ int32_t sh[256];
    const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + threadIdx.x * 16;
    int32_t v[16];
    int32_t s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { const int64_t i = base + k; v[k] = i < n ? in[i] : 0; s += v[k]; }
    int32_t run = offs[blockIdx.x] + block_exclusive_scan_256(s, sh, nullptr);
#pragma unroll
    for (int k = 0; k < 16; ++k) { const int64_t i = base + k; if (i < n) out[i] = run; run += v[k]; }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = offs[gridDim.x];
}

// out[0..n] = exclusive scan of in[0..n) (out[n] = total); scratch: 2*chunks + 1 ints
inline void launch_exclusive_scan(const int32_t* in, int32_t* out, int64_t n, int32_t* scratch, hipStream_t st) {
    if (n <= SCAN_CHUNK * 4) {        // small: one workgroup does it in one launch
        hipLaunchKernelGGL(exclusive_scan_kernel, dim3(1), dim3(1024), 0, st, in, out, n);
        return;
    }
    const int chunks = (int)((n + SCAN_CHUNK - 1) / SCAN_CHUNK);
    int32_t* sums = scratch;
    int32_t* offs = scratch + chunks;
    hipLaunchKernelGGL(scan_chunk_sums_kernel, dim3(chunks), dim3(256), 0, st, in, n, sums);
    hipLaunchKernelGGL(exclusive_scan_kernel, dim3(1), dim3(1024), 0, st, (const int32_t*)sums, offs, (int64_t)chunks);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(chunks), dim3(256), 0, st, in, n, (const int32_t*)offs, out);
}

__global__ void plan_tokens_kernel(const int32_t* __restrict__ sfm, int64_t n_rows, int seq, int pad, int lam,
                                   int n_ids, PlanArrays p) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_rows) return;
    const int32_t* row = sfm + n * seq;
    int t = p.row_offset[n];
    if (p.row_uniform[n] == 2) return;
    const bool uniform = p.row_uniform[n] == 1;
    const int lp = seq + lam;
    for (int j = 0; j < seq; ++j) {
        const int id = row[j];
        const bool vis = id != pad;
        if (uniform || vis || j == 0) {
            p.tok_slot[t] = p.id_slot[id];
            p.tok_pos[t] = j;
            p.tok_row[t] = (int32_t)n;
            p.tok_key[t] = vis ? 1 : 0;
            if (p.pair_flag) { const int key = id * lp + j; p.tok_pkey[t] = key; p.pair_flag[key] = 1; }
            ++t;
        }
    }
    if (lam) {
        p.tok_slot[t] = -1;
        p.tok_pos[t] = seq;
        p.tok_row[t] = (int32_t)n;
        p.tok_key[t] = 1;
        if (p.pair_flag) { const int key = n_ids * lp + seq; p.tok_pkey[t] = key; p.pair_flag[key] = 1; }
    }
}

// pair slot of every packed position, and (table slot, position) of every pair (tokens of one pair write the same values)
__global__ void plan_pairs_kernel(int64_t n_rows, PlanArrays p) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= p.row_offset[n_rows]) return;          // (the grid covers the worst case: the host does not know T yet)
    const int ps = p.pair_slot[p.tok_pkey[t]];
    p.tok_pair[t] = ps;
    p.pair_tslot[ps] = p.tok_slot[t];
    p.pair_pos[ps] = p.tok_pos[t];
}

// pair slot per BUFFER row of a chunk (position 0 first, chunk_row below): what the residual epilogue of layer 0 indexes with
__device__ __forceinline__ int chunk_row(int t_rel, int row_rel, bool first, int rows);
__global__ void pair_rows_kernel(int m, int tok0, int64_t row0, int rows, PlanArrays p, int32_t* __restrict__ brow_pair) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    const int n = p.tok_row[tok0 + t];
    brow_pair[chunk_row(t, (int)(n - row0), p.row_offset[n] == tok0 + t, rows)] = p.tok_pair[tok0 + t];
}

__global__ void plan_idlist_kernel(int n_ids, PlanArrays p) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id < n_ids && p.id_flag[id]) p.id_list[p.id_slot[id]] = id;
    if (id == 0) {          // (the last kernel of a plan: every scan has run)
        p.counters[0] = p.id_slot[n_ids];
        p.counters[1] = p.err[0];
        p.counters[2] = p.n_pair_keys ? p.pair_slot[p.n_pair_keys] : 0;
    }
}

// ---------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// sum over a 256-thread workgroup; every thread gets the result
__device__ __forceinline__ float block_sum_256(float v, float* red /* [4] */) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

template <typename T> __device__ __forceinline__ void store_lo4(T* dst, float4 v);
template <> __device__ __forceinline__ void store_lo4<float>(float* dst, float4 v) { *(float4*)dst = v; }
template <> __device__ __forceinline__ void store_lo4<bf16_t>(bf16_t* dst, float4 v) {
    *(uint2*)dst = make_uint2(pack2_lo<bf16_t>(v.x, v.y), pack2_lo<bf16_t>(v.z, v.w));
}
template <> __device__ __forceinline__ void store_lo4<f16_t>(f16_t* dst, float4 v) {
    *(uint2*)dst = make_uint2(pack2_lo<f16_t>(v.x, v.y), pack2_lo<f16_t>(v.z, v.w));
}

template <int SRC_DTYPE> __device__ __forceinline__ float4 load_src4(const void* base, size_t elem);
template <> __device__ __forceinline__ float4 load_src4<0>(const void* base, size_t e) {
    return *(const float4*)((const float*)base + e);
}
template <> __device__ __forceinline__ float4 load_src4<1>(const void* base, size_t e) {   // fp16
    const uint2 u = *(const uint2*)((const uint16_t*)base + e);
    typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
    const f16x4 hv = __builtin_bit_cast(f16x4, u);
    return make_float4((float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]);
}
template <> __device__ __forceinline__ float4 load_src4<2>(const void* base, size_t e) {   // bf16
    const uint2 u = *(const uint2*)((const uint16_t*)base + e);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                       __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}

// ---------------------------------------------------------------------------
// A2 + A3 (modeling_hypernet.py:170-188): one workgroup per distinct id.
//   id <  V0 : x = in_scaler.w * source_embeddings[id] + in_scaler.b   (if rescale)
//   id >= V0 : x = fallback_embeddings[id - V0]                         (never rescaled)
// ---------------------------------------------------------------------------
template <typename T, int SRC_DTYPE>
__global__ __launch_bounds__(256) void gather_src_kernel(const int32_t* __restrict__ id_list, int slot0, int n_slots,
                                                         const void* __restrict__ src, int e_in, int v0,
                                                         const float* __restrict__ fallback,
                                                         const float* __restrict__ sc_w, const float* __restrict__ sc_b,
                                                         T* __restrict__ out, int32_t* __restrict__ range_flag) {
    const int s = blockIdx.x;
    if (s >= n_slots) return;
    const int id = id_list[slot0 + s];
    T* dst = out + (size_t)s * e_in;
    bool bad = false;          // range guard: a value that does not fit the operand type (f16 only), ZETT_RANGE_SOURCE
    auto chk = [&](float4 v) {
        if (LoRange<T>::checked)
            bad |= out_of_range(v.x, LoRange<T>::limit) | out_of_range(v.y, LoRange<T>::limit) | out_of_range(v.z, LoRange<T>::limit) |
                   out_of_range(v.w, LoRange<T>::limit);
    };
    if (id >= v0) {
        const float* f = fallback + (size_t)(id - v0) * e_in;
        for (int c = threadIdx.x * 4; c < e_in; c += 1024) { const float4 v = *(const float4*)(f + c); chk(v); store_lo4<T>(dst + c, v); }
    } else {
        const size_t base = (size_t)id * e_in;
        for (int c = threadIdx.x * 4; c < e_in; c += 1024) {
            float4 v = load_src4<SRC_DTYPE>(src, base + c);
            if (sc_w) {
                const float4 w = *(const float4*)(sc_w + c), b = *(const float4*)(sc_b + c);
                v.x = w.x * v.x + b.x; v.y = w.y * v.y + b.y; v.z = w.z * v.z + b.z; v.w = w.w * v.w + b.w;
            }
            chk(v);
            store_lo4<T>(dst + c, v);
        }
    }
    range_report(range_flag, bad, ZETT_RANGE_BIT_SOURCE);
}

// ---------------------------------------------------------------------------
// LayerNorm over H for one row per workgroup (256 threads, float4 per thread, the row
// is held in registers for H <= 8192 and re-read otherwise).  Two-pass mean/variance
// like torch.nn.LayerNorm.
//   plain variant : x = in[row]
//   embed variant : RobertaEmbeddings (x + token_type[0] + position[pos]) where
//                   x = table[slot] or, for the language token (slot < 0),
//                   lang - (token_type[0] + position[L])   (modeling_hypernet.py:192-199)
// ---------------------------------------------------------------------------
struct LnEmbed {
    const float* table;        // [D, H] hoisted input projection
    const int32_t* tok_slot;
    const int32_t* tok_pos;
    const float* type0;        // [H]
    const float* pos_emb;      // [max_positions, H]
    const float* lang;         // [H] or null
    int lang_pos;              // L
    const int32_t* tok_row;    // [T] row of a packed position             } where the embed variant writes a position:
    const int32_t* row_offset; // [N+1]                                     } chunk_row() below
    int64_t row0;              // first vocabulary row of the chunk
    int rows;                  // vocabulary rows in the chunk
    // (r6) the table as the LayerNorm-fold producer left it: the 16-bit copy of the ProjectorBlock's PRE-LayerNorm sum, (mean, rstd)
    // per table row, and the block's gamma / beta — the embed variant normalises a table row as it reads it (ln_affine), so the
    // ProjectorBlock's LayerNorm is not a launch and a table element is 2 bytes instead of 4 on both sides.  null: `table` is fp32.
    const void* table_lo;      // [D, H] of the launch's operand type
    const float* table_stats;  // [D] (mean, rstd)
    const float* table_gamma;  // [H] input_projection.1.ln.weight
    const float* table_beta;   // [H]
};

// Row order of a chunk's hidden-state buffers: POSITION 0 FIRST.  A chunk covers vocabulary rows [row0, row0 + rows) =
// packed positions [tok0, tok0 + m) of the plan.  Buffer row r < rows holds position 0 of vocabulary row row0 + r; the other
// m - rows packed positions follow in plan order.  Only position 0 is read out (modeling_hypernet.py:234), so what the
// last layer and the output heads consume is then simply the first `rows` rows of every buffer — no gather in front of
// the last layer's query GEMM, none in front of the heads; GEMMs and LayerNorms do not care about the order of rows.
//   t_rel = packed position - tok0, row_rel = its vocabulary row - row0, first = it is the row's position 0
__device__ __forceinline__ int chunk_row(int t_rel, int row_rel, bool first, int rows) {
    return first ? row_rel : rows + t_rel - row_rel - 1;
}

// Position-0 readout fused into the LAST LayerNorm launch (which runs on the first `rows` buffer rows only): the bias
// head (modeling_hypernet.py:260-265) is a dot product of the fp32 LayerNorm output the kernel holds anyway.
struct LnReadout {
    const float* bias_w;       // [H] bias_projection.weight, or null (predict_bias off: the bias output is 0)
    const float* bias_b;       // [1]
    float* out_bias;           // [rows] (already offset to the chunk's first row), or null: no readout
    const void* in_lo;         // 16-bit residual stream: the rows come from here ([rows, ld_in] of the launch's operand type) instead of `in`
};

constexpr int LN_MAX_VEC = 8;   // 8 float4 x 256 threads = 8192 columns in registers

// TPR = threads per row: 256 (one workgroup per row, two LDS reductions) for wide rows, 64 (one wave per
// row, four rows per workgroup, shuffle reductions only) for H <= 2048, where a 256-thread group would
// leave lanes idle and spend its time in the two barriers.
// Outputs (each optional): out_lo = the normalised row as the 16-bit operand of the next GEMM; out_f32 = the
// normalised row in fp32 (hoisted table, output heads); stats_out = (mean, rstd) of the row, from which the residual
// epilogues and the position-0 readout recompute the fp32 row with ln_affine instead of reading it back;
// sum_out (embed variant) = the pre-LayerNorm sum itself, which those consumers then read.
// The embed variant takes packed position tok0 + r and writes buffer row chunk_row(...) (position 0 first).
// READOUT: the instantiation that also carries the bias head (LnReadout); the others ignore the argument.
template <typename T, bool EMBED, int TPR = 256, bool READOUT = false>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const float* __restrict__ in, int ld_in, int rows, int H,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps,
                                                             float* __restrict__ out_f32, T* __restrict__ out_lo,
                                                             float* __restrict__ stats_out, float* __restrict__ sum_out,
                                                             LnEmbed emb, int tok0, LnReadout readout) {
    __shared__ float red[4];
    const int r = blockIdx.x * (256 / TPR) + (int)threadIdx.x / TPR;
    if (r >= rows) return;                 // TPR = 64: whole waves leave, and no barrier follows
    const int nvec = H >> 2;
    const int tid = threadIdx.x % TPR;
    auto row_sum = [&](float x_) { return TPR == 256 ? block_sum_256(x_, red) : wave_sum(x_); };
    const float* x = nullptr;
    const float* posr = nullptr;
    bool is_lang = false;
    int ro = r;                            // output row
    const T* tbl_lo = nullptr;             // EMBED on the 16-bit table: the row's 16-bit pre-LayerNorm sum and its statistics
    float2 tbl_st = make_float2(0.f, 1.f);
    if constexpr (EMBED) {
        const int slot = emb.tok_slot[tok0 + r];
        is_lang = slot < 0;
        if (emb.table_lo && !is_lang) {
            tbl_lo = (const T*)emb.table_lo + (size_t)slot * H;
            tbl_st = *(const float2*)(emb.table_stats + 2 * (size_t)slot);
            x = emb.lang;                  // (not read)
        } else
        x = is_lang ? emb.lang : emb.table + (size_t)slot * H;
        posr = emb.pos_emb + (size_t)emb.tok_pos[tok0 + r] * H;
        if (emb.tok_row) {                 // (null: rows are pairs, written in place)
            const int n = emb.tok_row[tok0 + r];
            ro = chunk_row(r, (int)(n - emb.row0), emb.row_offset[n] == tok0 + r, emb.rows);
        }
    } else {
        x = in + (size_t)r * ld_in;
    }
    auto load = [&](int v) -> float4 {
        if constexpr (READOUT && sizeof(T) == 2) {
            if (readout.in_lo) {
                const uint2 u = *(const uint2*)((const T*)readout.in_lo + (size_t)r * ld_in + v * 4);
                float4 a;
                unpack2_lo<T>(u.x, a.x, a.y); unpack2_lo<T>(u.y, a.z, a.w);
                return a;
            }
        }
        float4 a;
        if constexpr (EMBED && sizeof(T) == 2) {
            if (tbl_lo) {          // (uniform per row: a table row of the 16-bit table, normalised on the fly)
                const uint2 u = *(const uint2*)(tbl_lo + v * 4);
                unpack2_lo<T>(u.x, a.x, a.y); unpack2_lo<T>(u.y, a.z, a.w);
                const float4 g = *(const float4*)(emb.table_gamma + v * 4), b = *(const float4*)(emb.table_beta + v * 4);
                a = make_float4(ln_affine(a.x, tbl_st.x, tbl_st.y, g.x, b.x), ln_affine(a.y, tbl_st.x, tbl_st.y, g.y, b.y),
                                ln_affine(a.z, tbl_st.x, tbl_st.y, g.z, b.z), ln_affine(a.w, tbl_st.x, tbl_st.y, g.w, b.w));
            } else {
                a = *(const float4*)(x + v * 4);
            }
        } else {
            a = *(const float4*)(x + v * 4);
        }
        if constexpr (EMBED) {
            const float4 t0 = *(const float4*)(emb.type0 + v * 4);
            const float4 p = *(const float4*)(posr + v * 4);
            if (is_lang) {   // lang -= type0 + pos[L]   (then the embeddings add them back)
                const float4 pl = *(const float4*)(emb.pos_emb + (size_t)emb.lang_pos * H + v * 4);
                a.x -= (t0.x + pl.x); a.y -= (t0.y + pl.y); a.z -= (t0.z + pl.z); a.w -= (t0.w + pl.w);
            }
            a.x = (a.x + t0.x) + p.x; a.y = (a.y + t0.y) + p.y; a.z = (a.z + t0.z) + p.z; a.w = (a.w + t0.w) + p.w;
        }
        return a;
    };
    float4 v[LN_MAX_VEC];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAX_VEC; ++j) {
        const int idx = tid + TPR * j;
        if (idx < nvec) { v[j] = load(idx); s += (v[j].x + v[j].y) + (v[j].z + v[j].w); }
    }
    for (int idx = tid + TPR * LN_MAX_VEC; idx < nvec; idx += TPR) { const float4 a = load(idx); s += (a.x + a.y) + (a.z + a.w); }
    const float mean = row_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAX_VEC; ++j) {
        const int idx = tid + TPR * j;
        if (idx < nvec) {
            const float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
    }
    for (int idx = tid + TPR * LN_MAX_VEC; idx < nvec; idx += TPR) {
        const float4 t = load(idx);
        const float a = t.x - mean, b = t.y - mean, c = t.z - mean, d = t.w - mean;
        q += (a * a + b * b) + (c * c + d * d);
    }
    const float var = row_sum(q) / (float)H;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (stats_out && tid == 0) *(float2*)(stats_out + 2 * (size_t)ro) = make_float2(mean, rstd);
    float dot = 0.f;
    auto emit = [&](int idx, float4 a) {
        const float4 g = *(const float4*)(gamma + idx * 4), b = *(const float4*)(beta + idx * 4);
        const float4 o = make_float4(ln_affine(a.x, mean, rstd, g.x, b.x), ln_affine(a.y, mean, rstd, g.y, b.y),
                                     ln_affine(a.z, mean, rstd, g.z, b.z), ln_affine(a.w, mean, rstd, g.w, b.w));
        if (EMBED && sum_out) *(float4*)(sum_out + (size_t)ro * H + idx * 4) = a;
        if (out_f32) *(float4*)(out_f32 + (size_t)ro * H + idx * 4) = o;
        if (out_lo) store_lo4<T>(out_lo + (size_t)ro * H + idx * 4, o);
        if (READOUT && readout.bias_w) {
            const float4 w = *(const float4*)(readout.bias_w + idx * 4);
            dot += (o.x * w.x + o.y * w.y) + (o.z * w.z + o.w * w.w);
        }
    };
#pragma unroll
    for (int j = 0; j < LN_MAX_VEC; ++j) {
        const int idx = tid + TPR * j;
        if (idx < nvec) emit(idx, v[j]);
    }
    for (int idx = tid + TPR * LN_MAX_VEC; idx < nvec; idx += TPR) emit(idx, load(idx));
    if constexpr (READOUT) {
        if (readout.out_bias) {            // (uniform over the launch: every thread of the row takes the reduction)
            const float tot = row_sum(dot);
            if (tid == 0) readout.out_bias[r] = readout.bias_w ? tot + readout.bias_b[0] : 0.f;
        }
    }
}

// ---------------------------------------------------------------------------
// (r2: keeping the <= 8 K/V vectors of a row in registers instead of re-reading them per query was measured and
//  rejected — 172 VGPRs, two waves per SIMD, 0.99 ms instead of 0.54 ms per launch: the re-reads hit L1/L2, and at
//  0.54 ms the kernel already moves its algorithmic 2.5 GB at 4.7 TB/s.)
// Attention over the packed positions of one row (A6).  One wave handles one row and
// one group of 512 hidden columns (8 per lane): 512/d heads at a time, head-local
// reductions by xor-shuffles over d/8 lanes.  Two passes over the keys (row max, then
// exp / sum / PV) reproduce softmax(QKᵀ·d^-½ + mask)·V of eager_attention_forward; keys
// with ids == pad contribute exactly 0 (their exp underflows to 0 in the reference),
// uniform rows weigh every position equally.
// ---------------------------------------------------------------------------
template <typename T> struct Vec8 { float v[8]; };

template <typename T> __device__ __forceinline__ void load8(const T* p, float (&o)[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&o)[8]) {
    const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
template <typename T> __device__ __forceinline__ void load8_16bit(const T* p, float (&o)[8]) {
    const uint4 u = *(const uint4*)p;
    unpack2_lo<T>(u.x, o[0], o[1]); unpack2_lo<T>(u.y, o[2], o[3]);
    unpack2_lo<T>(u.z, o[4], o[5]); unpack2_lo<T>(u.w, o[6], o[7]);
}
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float (&o)[8]) { load8_16bit<bf16_t>(p, o); }
template <> __device__ __forceinline__ void load8<f16_t>(const f16_t* p, float (&o)[8]) { load8_16bit<f16_t>(p, o); }
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&o)[8]);
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&o)[8]) {
    *(float4*)p = make_float4(o[0], o[1], o[2], o[3]);
    *(float4*)(p + 4) = make_float4(o[4], o[5], o[6], o[7]);
}
template <typename T> __device__ __forceinline__ void store8_16bit(T* p, const float (&o)[8]) {
    *(uint4*)p = make_uint4(pack2_lo<T>(o[0], o[1]), pack2_lo<T>(o[2], o[3]), pack2_lo<T>(o[4], o[5]), pack2_lo<T>(o[6], o[7]));
}
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float (&o)[8]) { store8_16bit<bf16_t>(p, o); }
template <> __device__ __forceinline__ void store8<f16_t>(f16_t* p, const float (&o)[8]) { store8_16bit<f16_t>(p, o); }

// ---- narrow rows (r6): layernorm_rows_kernel for 16-bit operand types and H <= 2048, EIGHT columns per lane -------------------
// The one-wave-per-row instantiation above (TPR = 64, float4 / 8-byte accesses) is latency-bound on the narrow hypernets: at
// H = 768 a wave has 1.5 KB in flight and four waves fit a SIMD (106 VGPRs) — 1.8 TB/s on XLM-R's embeddings launch.  Here a lane
// takes 8 consecutive columns (16-byte accesses on the 16-bit side, two float4 on the fp32 side) and a row takes TPR = 32 lanes
// for H <= 1024 (two rows per wave) or 64 for H <= 2048: H = 8 * TPR * NV with NV <= 4 vectors per lane.  Element for element the
// arithmetic of layernorm_rows_kernel (the same load / ln_affine / emit expressions); only the ORDER in which a row's values are
// added up for its mean and variance differs (8 per lane, then a butterfly over TPR lanes).
template <int TPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int off = TPR / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// lanes per row of layernorm_rows8_kernel for rows of H columns, or 0: the row does not fit (H = 8 * TPR * NV, NV <= 4)
inline int ln_rows8_tpr(int H) {
    if (H % 256 == 0 && H <= 1024) return 32;
    if (H % 512 == 0 && H <= 2048) return 64;
    return 0;
}

template <typename T, bool EMBED, int TPR, bool READOUT>
__global__ __launch_bounds__(256) void layernorm_rows8_kernel(const float* __restrict__ in, int ld_in, int rows, int H,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps,
                                                              float* __restrict__ out_f32, T* __restrict__ out_lo,
                                                              float* __restrict__ stats_out, float* __restrict__ sum_out,
                                                              LnEmbed emb, int tok0, LnReadout readout) {
    static_assert(sizeof(T) == 2 && (TPR == 32 || TPR == 64), "16-bit operand types, 32 or 64 lanes per row");
    constexpr int NVMAX = 4;
    const int r = blockIdx.x * (256 / TPR) + (int)threadIdx.x / TPR;
    if (r >= rows) return;                 // (a lane group leaves: the reductions below stay inside a group)
    const int nv = H / (8 * TPR);          // vectors of 8 columns per lane, <= NVMAX (the launcher checks)
    const int tid = threadIdx.x % TPR;
    const float* x = nullptr;
    const float* posr = nullptr;
    bool is_lang = false;
    int ro = r;
    const T* tbl_lo = nullptr;
    float2 tbl_st = make_float2(0.f, 1.f);
    if constexpr (EMBED) {
        const int slot = emb.tok_slot[tok0 + r];
        is_lang = slot < 0;
        if (emb.table_lo && !is_lang) {
            tbl_lo = (const T*)emb.table_lo + (size_t)slot * H;
            tbl_st = *(const float2*)(emb.table_stats + 2 * (size_t)slot);
            x = emb.lang;                  // (not read)
        } else
        x = is_lang ? emb.lang : emb.table + (size_t)slot * H;
        posr = emb.pos_emb + (size_t)emb.tok_pos[tok0 + r] * H;
        if (emb.tok_row) {
            const int n = emb.tok_row[tok0 + r];
            ro = chunk_row(r, (int)(n - emb.row0), emb.row_offset[n] == tok0 + r, emb.rows);
        }
    } else {
        x = in + (size_t)r * ld_in;
    }
    auto load = [&](int c, float (&a)[8]) {          // the 8 values of columns [c, c + 8)
        bool have = false;
        if constexpr (READOUT) {
            if (readout.in_lo) { load8<T>((const T*)readout.in_lo + (size_t)r * ld_in + c, a); have = true; }
        }
        if constexpr (EMBED) {
            if (tbl_lo) {          // a row of the 16-bit table, normalised on the fly
                load8<T>(tbl_lo + c, a);
                float g[8], b[8];
                load8<float>(emb.table_gamma + c, g);
                load8<float>(emb.table_beta + c, b);
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] = ln_affine(a[k], tbl_st.x, tbl_st.y, g[k], b[k]);
                have = true;
            }
        }
        if (!have) load8<float>(x + c, a);
        if constexpr (EMBED) {
            float t0[8], p[8];
            load8<float>(emb.type0 + c, t0);
            load8<float>(posr + c, p);
            if (is_lang) {   // lang -= type0 + pos[L]   (then the embeddings add them back)
                float pl[8];
                load8<float>(emb.pos_emb + (size_t)emb.lang_pos * H + c, pl);
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] -= (t0[k] + pl[k]);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = (a[k] + t0[k]) + p[k];
        }
    };
    float v[NVMAX][8];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NVMAX; ++j) {
        if (j < nv) {
            load((tid + TPR * j) * 8, v[j]);
            s += ((v[j][0] + v[j][1]) + (v[j][2] + v[j][3])) + ((v[j][4] + v[j][5]) + (v[j][6] + v[j][7]));
        }
    }
    const float mean = group_sum<TPR>(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NVMAX; ++j) {
        if (j < nv) {
            float d[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { d[k] = v[j][k] - mean; d[k] *= d[k]; }
            q += ((d[0] + d[1]) + (d[2] + d[3])) + ((d[4] + d[5]) + (d[6] + d[7]));
        }
    }
    const float var = group_sum<TPR>(q) / (float)H;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (stats_out && tid == 0) *(float2*)(stats_out + 2 * (size_t)ro) = make_float2(mean, rstd);
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < NVMAX; ++j) {
        if (j < nv) {
            const int c = (tid + TPR * j) * 8;
            float g[8], b[8], o[8];
            load8<float>(gamma + c, g);
            load8<float>(beta + c, b);
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = ln_affine(v[j][k], mean, rstd, g[k], b[k]);
            if (EMBED && sum_out) store8<float>(sum_out + (size_t)ro * H + c, v[j]);
            if (out_f32) store8<float>(out_f32 + (size_t)ro * H + c, o);
            if (out_lo) store8<T>(out_lo + (size_t)ro * H + c, o);
            if (READOUT && readout.bias_w) {
                float w[8];
                load8<float>(readout.bias_w + c, w);
                dot += ((o[0] * w[0] + o[1] * w[1]) + (o[2] * w[2] + o[3] * w[3])) + ((o[4] * w[4] + o[5] * w[5]) + (o[6] * w[6] + o[7] * w[7]));
            }
        }
    }
    if constexpr (READOUT) {
        if (readout.out_bias) {
            const float tot = group_sum<TPR>(dot);
            if (tid == 0) readout.out_bias[r] = readout.bias_w ? tot + readout.bias_b[0] : 0.f;
        }
    }
}

// ---- fast path of the attention kernel (16-bit operands, rows of at most ATT_FAST_KEYS packed positions: every row of the
// BASELINE workloads, hn_surface_maxlen 7 + language token) ----------------------------------------------------------------
// (r4) The generic loop below walks (query, key) pairs one after the other: a K and a V load, a shuffle reduction through the
// LDS crossbar and two expf chains per pair, each waiting for the one before — ~0.3 ns per pair and wave whatever the width,
// which on the narrow hypernets is 2-3 TB/s (XLM-R shape: 0.33 ms per launch for 1.04 GB).  Here the row's keys and values
// are fetched ONCE, all at the start (packed 16-bit, 8 registers per position), and a query handles all keys side by side:
// independent v_dot2 chains, head sums by DPP (no LDS), one v_exp_f32 per key, no running-maximum correction.
constexpr int ATT_FAST_KEYS = 8;

template <typename T> __device__ __forceinline__ float dot8_lo(const uint4& a, const uint4& b);
template <> __device__ __forceinline__ float dot8_lo<f16_t>(const uint4& a, const uint4& b) {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    float s = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, a.x), __builtin_bit_cast(h2, b.x), 0.f, false);
    s = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, a.y), __builtin_bit_cast(h2, b.y), s, false);
    s = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, a.z), __builtin_bit_cast(h2, b.z), s, false);
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, a.w), __builtin_bit_cast(h2, b.w), s, false);
}
template <> __device__ __forceinline__ float dot8_lo<bf16_t>(const uint4& a, const uint4& b) {
    float x[8], y[8];
    unpack2_lo<bf16_t>(a.x, x[0], x[1]); unpack2_lo<bf16_t>(a.y, x[2], x[3]); unpack2_lo<bf16_t>(a.z, x[4], x[5]); unpack2_lo<bf16_t>(a.w, x[6], x[7]);
    unpack2_lo<bf16_t>(b.x, y[0], y[1]); unpack2_lo<bf16_t>(b.y, y[2], y[3]); unpack2_lo<bf16_t>(b.z, y[4], y[5]); unpack2_lo<bf16_t>(b.w, y[6], y[7]);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) s = fmaf(x[c], y[c], s);
    return s;
}
template <> __device__ __forceinline__ float dot8_lo<float>(const uint4&, const uint4&) { return 0.f; }     // (never called: the fast path is 16-bit only)

// sum over the `lph` lanes of a head (lph = head_dim / 8, a power of two): DPP inside a row of 16 lanes, ds_swizzle (no LDS
// access) across 16, a bpermute only for a head that spans the whole wave.  Every lane of the head ends with the same bits.
__device__ __forceinline__ float head_sum(float s, int lph) {
    auto dpp = [](float v, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xf, 0xf, true));
    };
    if (lph >= 2) s += dpp(s, std::integral_constant<int, 0xB1>{});         // quad_perm [1,0,3,2]
    if (lph >= 4) s += dpp(s, std::integral_constant<int, 0x4E>{});         // quad_perm [2,3,0,1]
    if (lph >= 8) s += dpp(s, std::integral_constant<int, 0x141>{});        // row_half_mirror
    if (lph >= 16) s += dpp(s, std::integral_constant<int, 0x140>{});       // row_mirror
    if (lph >= 32) s += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, s), (16 << 10) | 0x1f));
    if (lph >= 64) s += __shfl_xor(s, 32, 64);
    return s;
}

// How the (row, 512-column group) work of the attention kernel maps to waves: `full` waves of one row and one whole group each
// (full_groups per row), then — `pack`, and the last group is 256 / 128 / 64 columns wide — waves that take that group of
// 64 / lanes_per_row rows at a time; otherwise the partial group is one more "full" wave per row with idle lanes.
struct AttentionWaves { int64_t total, full; int full_groups, lanes_per_row; };
__host__ __device__ inline AttentionWaves attention_waves(int rows, int H, bool pack) {
    const int rem = H & 511, gfull = H >> 9;
    AttentionWaves a;
    if (pack && (rem == 256 || rem == 128 || rem == 64)) {
        a.full_groups = gfull; a.lanes_per_row = rem >> 3;
        a.full = (int64_t)rows * gfull;
        const int per = 64 / a.lanes_per_row;
        a.total = a.full + (rows + per - 1) / per;
    } else {
        a.full_groups = (H + 511) >> 9; a.lanes_per_row = 64;
        a.full = (int64_t)rows * a.full_groups;
        a.total = a.full;
    }
    return a;
}

// Rows of q / k / v / ctx are BUFFER rows (position 0 first, chunk_row above); the plan arrays are indexed by packed position.
// q: [T, ldq] per packed position, or (cls_only) [rows, ldq] holding the query of position 0
// of each row; k, v: [T, ldkv]; ctx: [T or rows, H].  cls_only: compute query 0 only and
// write it at ctx[row_local] (compact [rows, H] output for the last layer).
template <typename T>
__global__ __launch_bounds__(256) void attention_rows_kernel(const T* __restrict__ qbase, size_t ldq,
                                                             const T* __restrict__ kbase, const T* __restrict__ vbase,
                                                             size_t ldkv, int H, int head_dim,
                                                             const int32_t* __restrict__ row_offset,
                                                             const uint8_t* __restrict__ row_uniform,
                                                             const uint8_t* __restrict__ tok_key, int64_t row0,
                                                             int rows, int tok0, float scaling, int flags /* bit 0: cls_only, bit 1: fast path on, bit 2: pack a narrow last group */,
                                                             const int32_t* __restrict__ tok_pair, T* __restrict__ ctx) {
    const int cls_only = flags & 1;
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    // (r6) a last column group of 256 / 128 / 64 columns (H = 768: the XLM-R shape) used to leave half or more of its wave's
    // lanes idle; with flag bit 2 such a wave takes that group of 2 / 4 / 8 ROWS instead (`lpr` lanes per row, `base` = the first
    // lane of this lane's row).  Everything per row below (t0, t1, the key mask, the pair slots) is then per lane group; heads
    // never straddle a group (the group's width is a multiple of head_dim), so the head sums stay inside it.  Same arithmetic
    // per element: same bits.
    const AttentionWaves aw = attention_waves(rows, H, (flags & 4) != 0);
    if (w >= aw.total) return;
    int rl, grp, li = lane, base = 0, lpr = 64;
    bool valid = true;
    if (w < aw.full) { rl = (int)(w / aw.full_groups); grp = (int)(w % aw.full_groups); }
    else {
        lpr = aw.lanes_per_row;
        const int sub = lane / lpr;
        li = lane - sub * lpr;
        base = sub * lpr;
        rl = (int)(w - aw.full) * (64 / lpr) + sub;
        grp = aw.full_groups;
        valid = rl < rows;
        if (!valid) rl = rows - 1;                     // (a lane group past the last row repeats it and stores nothing)
    }
    const int col = grp * 512 + li * 8;
    const bool active = valid && col < H;
    const int lph = head_dim >> 3;                     // lanes per head (power of two)
    const int64_t n = row0 + rl;
    const int t0 = row_offset[n] - tok0, t1 = row_offset[n + 1] - tok0;
    const bool uniform = row_uniform[n];
    const int nq = cls_only ? 1 : (t1 - t0);
    const int ccol = active ? col : 0;
    // buffer row of packed position t of this row (position 0 first: chunk_row above)
    auto brow = [&](int t) -> size_t { return (size_t)chunk_row(t, rl, t == t0, rows); };
    // row of q / k / v of packed position t: its buffer row, or — layer 0 with the pair lever — the row of its
    // (source id, position) pair, whose Q/K/V were computed once
    // (the row's first 64 pair slots are fetched once, one per lane, and handed out by shuffle: an index load in front of
    //  every key would put a second memory round trip into the loop; a row with more than 64 packed positions —
    //  hn_surface_maxlen >= 64 is legal, max_positions is 514 — reads the slots past the 64th directly: t is
    //  wave-uniform, so the branch is too)
    int pslot = 0;
    if (tok_pair) { const int t = t0 + li; pslot = t < t1 ? tok_pair[tok0 + t] : 0; }
    auto qrow = [&](int t) -> size_t {
        if (!tok_pair) return brow(t);
        return t - t0 < lpr ? (size_t)__shfl(pslot, base + (t - t0), 64) : (size_t)tok_pair[tok0 + t];
    };
    const int nk = t1 - t0;
    if constexpr (sizeof(T) == 2) {
        if (nk <= ATT_FAST_KEYS && (flags & 2)) {
            // which keys count: lane j asks for key j, the answer is a wave-uniform bit mask
            const bool mine = li < nk && (uniform || tok_key[tok0 + t0 + li] != 0);
            const unsigned on = (unsigned)(__ballot(mine) >> base) & (lpr >= 32 ? 0xffffffffu : ((1u << lpr) - 1u));
            uint4 kk[ATT_FAST_KEYS], vv[ATT_FAST_KEYS], qq[ATT_FAST_KEYS];
#pragma unroll
            for (int j = 0; j < ATT_FAST_KEYS; ++j) {
                if (j < nk) {
                    const size_t kr = qrow(t0 + j);
                    kk[j] = *(const uint4*)(kbase + kr * ldkv + ccol);
                    vv[j] = *(const uint4*)(vbase + kr * ldkv + ccol);
                    if (!cls_only) qq[j] = *(const uint4*)(qbase + kr * ldq + ccol);
                }
            }
            if (cls_only) qq[0] = *(const uint4*)(qbase + (size_t)rl * ldq + ccol);
            const float sc2 = scaling * 1.4426950408889634f;       // softmax in base 2: exp(x) = 2^(x log2 e)
#pragma unroll
            for (int qi = 0; qi < ATT_FAST_KEYS; ++qi) {
                if (qi < nq) {
                    float sj[ATT_FAST_KEYS];
                    float mx = -INFINITY;
#pragma unroll
                    for (int j = 0; j < ATT_FAST_KEYS; ++j) {
                        sj[j] = -INFINITY;
                        if (j < nk && ((on >> j) & 1u)) {
                            const float d = head_sum(dot8_lo<T>(qq[qi], kk[j]), lph);
                            sj[j] = uniform ? 0.f : d * sc2;
                            mx = fmaxf(mx, sj[j]);
                        }
                    }
                    float l = 0.f, acc[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
#pragma unroll
                    for (int j = 0; j < ATT_FAST_KEYS; ++j) {
                        if (j < nk && ((on >> j) & 1u)) {
                            const float pj = __builtin_amdgcn_exp2f(sj[j] - mx);
                            l += pj;
                            float v[8];
                            unpack2_lo<T>(vv[j].x, v[0], v[1]); unpack2_lo<T>(vv[j].y, v[2], v[3]);
                            unpack2_lo<T>(vv[j].z, v[4], v[5]); unpack2_lo<T>(vv[j].w, v[6], v[7]);
#pragma unroll
                            for (int c = 0; c < 8; ++c) acc[c] = fmaf(pj, v[c], acc[c]);
                        }
                    }
                    const float inv = 1.0f / l;
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc[c] *= inv;
                    if (active) {
                        const size_t orow = cls_only ? (size_t)rl : brow(t0 + qi);
                        store8<T>(ctx + orow * H + col, acc);
                    }
                }
            }
            return;
        }
    }
    for (int qi = 0; qi < nq; ++qi) {
        float q[8];
        load8<T>(qbase + (cls_only ? (size_t)rl : qrow(t0 + qi)) * ldq + ccol, q);
        // one pass over the keys with a running maximum (online softmax): K and V are read once
        float mx = -INFINITY, l = 0.f, acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = 0.f;
        for (int kj = t0; kj < t1; ++kj) {
            if (!uniform && !tok_key[tok0 + kj]) continue;
            float k[8], v[8];
            const size_t kr = qrow(kj);
            load8<T>(kbase + kr * ldkv + ccol, k);
            load8<T>(vbase + kr * ldkv + ccol, v);
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) s = fmaf(q[c], k[c], s);
            for (int off = 1; off < lph; off <<= 1) s += __shfl_xor(s, off, 64);
            s = uniform ? 0.f : s * scaling;
            const float nm = fmaxf(mx, s);
            const float corr = expf(mx - nm);          // first key: exp(-inf) = 0 on an all-zero state
            const float p = expf(s - nm);
            l = fmaf(l, corr, p);
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] = fmaf(acc[c], corr, p * v[c]);
            mx = nm;
        }
        const float inv = 1.0f / l;
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] *= inv;
        if (active) {
            const size_t orow = cls_only ? (size_t)rl : brow(t0 + qi);
            store8<T>(ctx + orow * H + col, acc);
        }
    }
}

// ---------------------------------------------------------------------------
// LayerNorm fold: the LayerNorm between two GEMMs without a kernel that reads the fp32 rows again.
//   ln_stats_kernel   (mean, rstd) per row from the per-128-column partials the producer GEMM's epilogue wrote
//   fold_weight_kernel  one-off, at zett_finalize: W'[n,k] = lo(W[n,k] * gamma[k]),  c[n] = sum_k W'[n,k],
//                       b'[n] = b[n] + sum_k W[n,k] * beta[k]     so that   LN(x) W^T + b = rstd (x W'^T - mean c) + b'
// ---------------------------------------------------------------------------
// (r6: a version with four threads per row — every fourth partial each, combined through LDS — was 2-3 x faster on a 12-16 us kernel
//  and changed the summation order: bf16 mode, which sits ON its 1e-2 tolerance, went from 0.99e-2 to 1.002e-2 on the Llama-3 sample.
//  The sequential order of rounds 3-5 stays.)
// (r6, second version: the partials of a row are still ADDED in their sequential order — same bits — but FETCHED eight at a time:
//  the loop of rounds 3-5 waited for every load before it issued the next, 32 dependent round trips at H = 4096.  64-thread
//  workgroups, so that a 4 096-row shard's 9.7 k rows reach 150 CUs instead of 38.)
__global__ __launch_bounds__(64) void ln_stats_kernel(const float2* __restrict__ part, int parts, int ld_part, int rows, int H, float eps,
                                                      float* __restrict__ stats) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    float s = 0.f, q = 0.f;
    const float2* col = part + r;
    int p = 0;
    for (; p + 8 <= parts; p += 8) {
        float2 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = col[(size_t)(p + j) * ld_part];
#pragma unroll
        for (int j = 0; j < 8; ++j) { s += v[j].x; q += v[j].y; }
    }
    {
        float2 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p + j < parts ? col[(size_t)(p + j) * ld_part] : make_float2(0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (p + j < parts) { s += v[j].x; q += v[j].y; }
    }
    const float mean = s / (float)H;
    const float var = fmaxf(q / (float)H - mean * mean, 0.f);
    *(float2*)(stats + 2 * (size_t)r) = make_float2(mean, 1.0f / sqrtf(var + eps));
}

template <typename T>
__global__ __launch_bounds__(256) void fold_weight_kernel(const float* __restrict__ w, int K, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ bias,
                                                          T* __restrict__ w_fold, float* __restrict__ c_out, float* __restrict__ b_out,
                                                          int32_t* __restrict__ range_flag) {
    __shared__ float red[4];
    const int n = blockIdx.x;
    const float* row = w + (size_t)n * K;
    float cs = 0.f, bs = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) {
        const float x = row[k];
        if (LoRange<T>::checked) range_report(range_flag, out_of_range(x * gamma[k], LoRange<T>::limit), ZETT_RANGE_BIT_WEIGHT);
        const T lo = to_lo<T>(x * gamma[k]);
        w_fold[(size_t)n * K + k] = lo;
        cs += lo_to_f32<T>(lo);
        bs = fmaf(x, beta[k], bs);
    }
    const float ct = block_sum_256(cs, red);
    const float bt = block_sum_256(bs, red);
    if (threadIdx.x == 0) { c_out[n] = ct; b_out[n] = bias[n] + bt; }
}

// dtype conversion used when weights are uploaded
template <typename T>
__global__ void convert_f32_to_lo_kernel(const float* __restrict__ in, T* __restrict__ out, size_t n, int32_t* __restrict__ range_flag) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    bool bad = false;          // range guard: a weight that does not fit the operand type (f16 only), ZETT_RANGE_WEIGHT
    for (; i + 3 < n; i += stride) {
        const float4 v = *(const float4*)(in + i);
        if (LoRange<T>::checked)
            bad |= out_of_range(v.x, LoRange<T>::limit) | out_of_range(v.y, LoRange<T>::limit) | out_of_range(v.z, LoRange<T>::limit) |
                   out_of_range(v.w, LoRange<T>::limit);
        store_lo4<T>(out + i, v);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (size_t j = n & ~(size_t)3; j < n; ++j) { out[j] = to_lo<T>(in[j]); if (LoRange<T>::checked) bad |= out_of_range(in[j], LoRange<T>::limit); }
    }
    range_report(range_flag, bad, ZETT_RANGE_BIT_WEIGHT);
}
template <int SRC_DTYPE>
__global__ void convert_to_f32_kernel(const void* __restrict__ in, float* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        if constexpr (SRC_DTYPE == 1) out[i] = (float)((const _Float16*)in)[i];
        else out[i] = __uint_as_float(((uint32_t)((const uint16_t*)in)[i]) << 16);
    }
}

}  // namespace zett
