// zett_hip.hip — C ABI of libzett_hip.so (see include/zett_hip.h).
//
// Host-side orchestration of the hypernet forward on one MI355X:
//
//   plan      surface forms -> packed positions, distinct referenced source ids
//   table     for each DISTINCT source id once: gather + in_scaler/fallback ->
//             input_projection (Linear + ProjectorBlock)            [hoisting, exact]
//   encoder   RobertaEmbeddings + hn_n_layers encoder layers over the PACKED
//             positions only (pad positions never influence hidden[:,0])  [exact]
//             last layer: keys/values for every position, query / O-proj / FFN for
//             position 0 only                                            [exact]
//   heads     ProjectorBlock + Linear (+ Rescaler) per output head, bias head
//
// Every dense contraction goes through gemm.hip.h (MFMA); everything else through
// the streaming kernels of rowops.hip.h.  All launches go to the caller's stream.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cstdlib>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/zett_hip.h"
#include "common.hip.h"
#include "gemm.hip.h"
#include "gemm_launch.hip.h"
#include "rowops.hip.h"
#include "retok.hip.h"
#include "partition.hip.h"

using namespace zett;

namespace {

struct Tensor {
    float* f32 = nullptr;       // device fp32 copy (biases, LN, embeddings, ... and GEMM weights in F32 mode)
    void* lo = nullptr;         // GEMM operand copy in the handle's arithmetic type
    std::vector<int64_t> shape;
    size_t numel = 0;
};

inline size_t round_up(size_t v, size_t m) { return (v + m - 1) / m * m; }

}  // namespace

struct zett_hypernet {
    zett_config cfg{};
    int device = 0;
    int precision = ZETT_PREC_BF16;
    bool finalized = false;
    std::map<std::string, Tensor> w;
    // fused / derived operands built by zett_finalize
    std::vector<void*> qkv_w;         // per layer [3H, H]
    std::vector<float*> qkv_b;        // per layer [3H]
    // LayerNorm fold (16-bit modes): gamma-folded operands of the GEMMs that follow an encoder LayerNorm — the fused QKV of
    // layers >= 1 (folded with the previous layer's output LayerNorm) and every intermediate.dense (with the layer's
    // attention-output LayerNorm) — with their row sums and beta-folded biases
    struct Folded { void* w = nullptr; float* c = nullptr; float* b = nullptr; };
    std::vector<Folded> fold_qkv, fold_up;
    Folded fold_head_in, fold_head_out;    // the final Linear of each output head, folded with its ProjectorBlock's LayerNorm
    float* head_one = nullptr;        // [head_out_width] ones / zeros: the Rescaler of a config without hn_rescale_embeddings, for the
    float* head_zero = nullptr;       // folded head epilogue, which is compiled with it
    int ln_fold = 1;
    float* head_scale = nullptr;      // [head_out_width] scaler.w (| out_scaler.w for single_head)
    float* head_shift = nullptr;
    std::vector<void*> owned;         // everything to hipFree at destroy
    // options
    int64_t max_chunk_tokens = 0;          // 0 = auto: chunk_token_cap() below (131 072 at H = 4096, more for narrower hypernets)
    int time_gemm = 0;
    int cls_only_last = 1;
    int pair_dedupe = 1;              // layer 0's Q/K/V once per distinct (source id, position) pair (do_forward)
    int residual_lo = 1;              // 16-bit residual stream of the encoder (gemm4d LN16 producers): 1 = in f16 mode, 2 = in bf16 mode too (A/B), 0 = fp32 stream
    int concurrent_lanes = 0;         // a one-chunk call as TWO half-vocabulary chunks on two streams (do_forward): 0 never, 1 auto (when the narrow GEMMs
                                      // of the call leave > 8 % of their last round of 256 CUs idle: the 4 096-row shards of 8 GPUs), 2 always
    hipStream_t lane_stream = nullptr;
    hipEvent_t lane_ev[4] = {nullptr, nullptr, nullptr, nullptr};      // fork, lane 1: bias written / out_in written / done
    int attention_fast = 1;           // rows of <= 8 packed positions: keys / values fetched once, all keys of a query side by side (rowops.hip.h)
    int attention_pack = 1;           // a last column group of 256 / 128 / 64 columns (H = 768) takes 2 / 4 / 8 rows per wave instead of idle lanes (r6; same bits)
    int gemm_tail_split = 1;          // gemm4d: a launch's partly filled last round of 256 CUs as 128x256 tiles (gemm4d.hip.h gemm4d_row_split): 0 never,
                                      // 1 when the split is cheaper (default), 2 = cut every launch in the middle, 3 = half tiles only (tests: same bits)
    int gemm_tile_order = 0;          // gemm4d: 0 = column-tile-major groups, 1 = row-tile-major groups (A/B)
    int ln_rows8 = 1;                 // 16-bit modes, H <= 2048: LayerNorm launches on layernorm_rows8_kernel (eight columns per lane; rowops.hip.h); 0 = the float4 kernel (A/B)
    int table_lo = 1;                 // 16-bit hoisted table with the ProjectorBlock's LayerNorm folded into the embeddings' kernel (with the 16-bit residual stream); 0 = fp32 table (A/B)
    int gemm_group = 0;               // gemm4d: column (order 0) / row (order 1) tiles per group; 0 = the kernel's default, 4 (A/B)
    int gemm4d_min_k = 512;           // 16-bit launches with K >= this take the four-wave direct-to-LDS tile (r2: with the streamlined epilogues it is ahead of gemm8r down to K = 768: +1.8 % on the XLM-R workload)
    int gemm_variant = 0;             // 0 auto, 1 = 128x128, 2 = 256x256 register-staged (8 waves), 3 = 384x256 LDS-DMA,
                                      // 7 = 256x256 four-wave direct-to-LDS, 8 = as 7 with the generic epilogue drain
    // workspace
    // The plan of a forward (integers: packed positions, distinct ids, pairs) lives in one of TWO slots that forwards take in
    // turn, so that the plan of the NEXT forward can be made (zett_forward_prepare, on plan_stream) while the kernels of the
    // current one still read theirs.
    struct PlanSlot {
        DevBuf i32, u8;
        int32_t* host = nullptr;          // pinned: row offsets [N+1], distinct ids, error word, distinct pairs
        size_t host_ints = 0;
        hipEvent_t done = nullptr;        // the plan's kernels and copies have run
        hipEvent_t released = nullptr;    // the forward that used this slot has finished with it
        const int32_t* sfm = nullptr;     // what the slot was prepared for (zett_forward_prepare), pending until a forward takes it
        int64_t n_rows = 0;
        int seq = 0;
        bool pair_plan = false;           // the layout it was made with (an option changed in between: the forward plans again)
        const int32_t* ext_id_slot = nullptr;   // the id -> table-slot map its tok_slot was written with (null: the plan's own)
        bool pending = false;
    } plan[2];
    PlanSlot table_plan;              // (ABI 8) scratch plan of zett_table_plan: the distinct ids of a WHOLE vocabulary, never taken by a forward
    // (ABI 8) the hoisted table of the forward in flight comes from the caller (zett_forward_table): rows of the global distinct-id
    // list computed by zett_table_rows — on this rank and on its peers — instead of this call's own distinct ids
    struct ExtTable { const void* table = nullptr; const float* stats = nullptr; const int32_t* id_slot = nullptr; } ext;
    int plan_cur = 0;                 // slot of the most recent forward
    hipStream_t plan_stream = nullptr;
    hipEvent_t plan_fork = nullptr;
    DevBuf table, x0, yf, yt, big, pre, ctx, cf, ct, lnstats, lnparts;
    hipEvent_t out_ready[2] = {nullptr, nullptr};      // zett_stream_wait_output: out_in / out_bias of the last forward complete
    bool out_recorded = false;
    int range_accumulate = 0;         // 1: zett_forward does not clear the range word, zett_check_range clears it after reading (a caller
                                      // that runs several asynchronous forwards and asks once at the end: zett_amd/sharding.py)
    int32_t* range_word = nullptr;    // device: zett_range_bits of the forward in flight (cleared when a forward starts)
    int32_t* range_host = nullptr;    // pinned: where zett_check_range / zett_finalize read it
    std::vector<hipEvent_t> ev;
    size_t ev_used = 0;
    std::vector<double> ev_flops;
    std::vector<std::array<int, 4>> ev_shape;   // M, N, K, variant of each timed launch
    std::vector<zett_gemm_record> gemm_log;     // every GEMM launch of the last forward (zett_get_gemm_log)
    zett_stats stats{};
};

namespace {

size_t elt_size(int precision) { return precision == ZETT_PREC_F32 ? 4 : 2; }

bool is_gemm_weight(const std::string& n) {
    static const char* suffixes[] = {"input_projection.0.weight", "dense1.weight", "dense2.weight",
                                     "query.weight", "key.weight", "value.weight", "dense.weight",
                                     "output_projection.1.weight", "output_projection_out.1.weight"};
    for (const char* s : suffixes) {
        const size_t ls = strlen(s);
        if (n.size() >= ls && n.compare(n.size() - ls, ls, s) == 0) return true;
    }
    return false;
}

// expected shapes (mirrors zett_amd/dims.py:weight_shapes)
void expected_shapes(const zett_config& c, std::map<std::string, std::vector<int64_t>>& s) {
    const int64_t H = c.hidden, I = c.intermediate;
    auto proj = [&](const std::string& p) {
        s[p + "dense1.weight"] = {I, H}; s[p + "dense1.bias"] = {I};
        s[p + "dense2.weight"] = {H, I}; s[p + "dense2.bias"] = {H};
        s[p + "ln.weight"] = {H}; s[p + "ln.bias"] = {H};
    };
    if (c.embed_lang) s["lang_embeddings.weight"] = {c.n_langs, H};
    s["model.embeddings.token_type_embeddings.weight"] = {1, H};
    s["model.embeddings.LayerNorm.weight"] = {H};
    s["model.embeddings.LayerNorm.bias"] = {H};
    s["model.embeddings.position_embeddings.weight"] = {c.max_positions, H};
    for (int l = 0; l < c.layers; ++l) {
        const std::string p = "model.encoder.layer." + std::to_string(l) + ".";
        for (const char* q : {"query", "key", "value"}) {
            s[p + "attention.self." + q + ".weight"] = {H, H};
            s[p + "attention.self." + q + ".bias"] = {H};
        }
        s[p + "attention.output.dense.weight"] = {H, H}; s[p + "attention.output.dense.bias"] = {H};
        s[p + "attention.output.LayerNorm.weight"] = {H}; s[p + "attention.output.LayerNorm.bias"] = {H};
        s[p + "intermediate.dense.weight"] = {I, H}; s[p + "intermediate.dense.bias"] = {I};
        s[p + "output.dense.weight"] = {H, I}; s[p + "output.dense.bias"] = {H};
        s[p + "output.LayerNorm.weight"] = {H}; s[p + "output.LayerNorm.bias"] = {H};
    }
    s["fallback_embeddings.weight"] = {c.n_extra, c.n_in_embd};
    s["input_projection.0.weight"] = {H, c.n_in_embd}; s["input_projection.0.bias"] = {H};
    proj("input_projection.1.");
    proj("output_projection.0.");
    const int64_t w0 = c.single_head ? c.n_in_embd : c.n_embd;
    s["output_projection.1.weight"] = {w0, H}; s["output_projection.1.bias"] = {w0};
    if (c.separate_out && !c.single_head) {
        proj("output_projection_out.0.");
        s["output_projection_out.1.weight"] = {c.n_embd, H}; s["output_projection_out.1.bias"] = {c.n_embd};
    }
    if (c.rescale) {
        s["in_scaler.w"] = {1, c.n_in_embd}; s["in_scaler.b"] = {1, c.n_in_embd};
        s["scaler.w"] = {1, c.n_embd}; s["scaler.b"] = {1, c.n_embd};
        if (c.separate_out) { s["out_scaler.w"] = {1, c.n_embd}; s["out_scaler.b"] = {1, c.n_embd}; }
    }
    if (c.predict_bias) { s["bias_projection.weight"] = {1, H}; s["bias_projection.bias"] = {1}; }
}

int validate_config(const zett_config& c, int precision) {
    if (c.n_embd <= 0 || c.hidden <= 0 || c.intermediate <= 0 || c.heads <= 0 || c.layers <= 0)
        return fail(ZETT_E_INVALID, "non-positive dimension in zett_config");
    if (c.n_in_embd != (c.separate_out ? 2 * c.n_embd : c.n_embd))
        return fail(ZETT_E_INVALID, "n_in_embd must be 2*n_embd iff separate_out");
    if (c.hidden % c.heads) return fail(ZETT_E_INVALID, "hidden %d not divisible by heads %d", c.hidden, c.heads);
    const int d = c.hidden / c.heads;
    if (d < 8 || d > 512 || (d & (d - 1))) return fail(ZETT_E_INVALID, "head_dim %d unsupported (need a power of two in [8,512])", d);
    const int kq = precision == ZETT_PREC_F32 ? 32 : 64;
    for (int k : {c.n_in_embd, c.hidden, c.intermediate})
        if (k % kq) return fail(ZETT_E_INVALID, "contraction width %d is not a multiple of %d", k, kq);
    if (c.n_embd % 4) return fail(ZETT_E_INVALID, "n_embd must be a multiple of 4");
    if (c.n_extra < 1) return fail(ZETT_E_INVALID, "n_extra must be >= 1 (reference allocates max(hn_n_extra_tokens,1))");
    if (c.pad_token_id < 0) return fail(ZETT_E_INVALID, "pad_token_id is required");
    if (c.embed_lang && c.n_langs <= 0) return fail(ZETT_E_INVALID, "hn_embed_lang_id needs n_langs");
    return 0;
}

// Device bytes zett_forward reserves, as a function of what the plan found (packed positions
// Ttot, distinct ids D).  One definition shared by do_forward and zett_workspace_bytes.
struct WorkspaceSizes {
    int64_t chunk_tokens;
    size_t table, x0, f32_rows, lo_rows, big, stats;
    size_t total() const { return table + x0 + 3 * f32_rows + 3 * lo_rows + big + stats; }
};

static int64_t pair_keys(const zett_config& c, int seq) {
    return ((int64_t)c.original_vocab_size + c.n_extra + 1) * (int64_t)(seq + (c.embed_lang ? 1 : 0));
}

static size_t plan_i32_bytes(const zett_config& c, int64_t N, int seq) {
    const int64_t V = (int64_t)c.original_vocab_size + c.n_extra;
    const int64_t max_tok = N * (int64_t)(seq + (c.embed_lang ? 1 : 0));
    // row_count[N] row_offset[N+1] id_flag[V] id_slot[V+1] id_list[V] tok_slot[T] tok_pos[T] tok_row[T] err[1] scan scratch
    // + the pair plan: tok_pkey[T] tok_pair[T] pair_tslot[T] pair_pos[T] pair_flag[K] pair_slot[K+1], K = (V+1)(L+lang)
    const int64_t K = pair_keys(c, seq);
    const size_t scan_scratch = 2 * ((size_t)std::max<int64_t>(std::max<int64_t>(N, V), K) / SCAN_CHUNK + 2) + 1;
    return ((size_t)N + (N + 1) + 3 + V + (V + 1) + V + 7 * (size_t)max_tok + 2 * (size_t)K + 2 + scan_scratch) * 4;
}

// Packed positions per encoder chunk when the caller has not set "max_chunk_tokens": 12 GiB of per-position workspace,
// never under 131 072 positions.  That is 131 072 at H = 4096 (112 KB per position), 230 k at H = 2048, 640 k at
// H = 768: a narrow hypernet's vocabulary stays ONE chunk (the 50 350-row XLM-R workload packs 169 283 positions; cut at
// 131 072 it ran a second, 38 211-position chunk whose GEMMs and ~35 extra launches cost 5 % of the step).
static int64_t chunk_token_cap(const zett_config& c, size_t es) {
    const size_t wide = (size_t)std::max(c.intermediate, 3 * c.hidden);
    const size_t per_token = (size_t)c.n_in_embd * es + 3 * (size_t)c.hidden * 4 + 3 * (size_t)c.hidden * es + wide * es + 24;
    return std::max<int64_t>(131072, (int64_t)(((size_t)12 << 30) / per_token));
}

static WorkspaceSizes workspace_sizes(const zett_config& c, size_t es, int seq, int64_t Ttot, int64_t D, int64_t cap) {
    const int lam = c.embed_lang ? 1 : 0;
    WorkspaceSizes w{};
    if (cap <= 0) cap = chunk_token_cap(c, es);
    w.chunk_tokens = std::max<int64_t>(std::min<int64_t>(cap, std::max<int64_t>(Ttot, D)), seq + lam);
    const size_t MC = (size_t)w.chunk_tokens, MCS = MC + 768;       // (384 slack rows per lane: two concurrent lanes at most)
    const size_t wide = (size_t)std::max(c.intermediate, 3 * c.hidden);
    w.table = (size_t)D * c.hidden * 4;
    w.x0 = MCS * c.n_in_embd * es;
    w.f32_rows = MCS * c.hidden * 4;               // (the second lane's slice starts 384 rows behind the first one's rows)
    w.lo_rows = MCS * c.hidden * es;
    w.big = MCS * wide * es;
    w.stats = 2 * MCS * 2 * sizeof(float);         // (mean, rstd) per row: two LayerNorms in flight
    return w;
}

// Where the arrays of a plan lie in a slot (pure pointer arithmetic: the same answer when the plan is made and when a
// forward takes it).  Pair plan (lever 4): only where it can pay — at least two encoder layers (layer 0 must not be the
// position-0-only one), and a key space not much larger than the batch (a 250 k-id source vocabulary against 50 k rows
// repeats few pairs and would pay for clearing and scanning 2 M flags).
struct PlanLayout {
    PlanArrays p{};
    int32_t* scan_tmp = nullptr;
    bool pair_plan = false;
    int64_t PK = 0;
    size_t n_clear = 0;          // int32 words at the head of the arena that a plan starts from zero
};
static PlanLayout plan_layout(const zett_hypernet* h, const zett_hypernet::PlanSlot& s, int64_t N, int seq) {
    const zett_config& c = h->cfg;
    const int lam = c.embed_lang ? 1 : 0;
    const int V = c.original_vocab_size + c.n_extra;
    const int64_t max_tok = N * (int64_t)(seq + lam);
    PlanLayout L;
    PlanArrays& p = L.p;
    // int32 arena: what a plan clears first, in one piece — id_flag[V] pair_flag[PK] err[1] — then row_count[N] row_offset[N+1]
    // counters[3] (read by the host in one copy with the row offsets) id_slot[V+1] id_list[V] tok_slot[T] tok_pos[T] tok_row[T]
    // [pair plan] scan scratch
    L.PK = pair_keys(c, seq);
    L.pair_plan = h->pair_dedupe && c.layers >= 2 && L.PK <= std::max<int64_t>(4 * max_tok, (int64_t)1 << 23) && L.PK < (int64_t)0x7fffffff;
    int32_t* base = s.i32.as<int32_t>();
    p.id_flag = base; base += V;
    if (L.pair_plan) { p.pair_flag = base; base += L.PK; }
    p.err = base; base += 1;
    L.n_clear = (size_t)(base - s.i32.as<int32_t>());
    p.row_count = base; base += N;
    p.row_offset = base; base += N + 1;
    p.counters = base; base += 3;
    p.id_slot = base; base += V + 1;
    p.id_list = base; base += V;
    p.tok_slot = base; base += max_tok;
    p.tok_pos = base; base += max_tok;
    p.tok_row = base; base += max_tok;
    p.n_pair_keys = L.pair_plan ? (int32_t)L.PK : 0;
    if (L.pair_plan) {
        p.tok_pkey = base; base += max_tok;
        p.tok_pair = base; base += max_tok;
        p.pair_tslot = base; base += max_tok;
        p.pair_pos = base; base += max_tok;
        p.pair_slot = base; base += L.PK + 1;
    }
    L.scan_tmp = base;
    p.row_uniform = s.u8.as<uint8_t>();
    p.tok_key = p.row_uniform + N;
    return L;
}

// The plan of one forward into slot s, on stream st: six small kernels, three scans, and the copy of the row offsets and the
// three counters to pinned memory; s.done is recorded behind them.
static int enqueue_plan(zett_hypernet* h, zett_hypernet::PlanSlot& s, const int32_t* sfm, int64_t N, int seq, hipStream_t st) {
    const zett_config& c = h->cfg;
    const int lam = c.embed_lang ? 1 : 0;
    const int V = c.original_vocab_size + c.n_extra;
    const int64_t max_tok = N * (int64_t)(seq + lam);
    if (int rc = s.i32.reserve(plan_i32_bytes(c, N, seq))) return rc;
    if (int rc = s.u8.reserve((size_t)N + (size_t)max_tok)) return rc;
    const size_t need_ints = (size_t)N + 1 + 3;
    if (s.host_ints < need_ints) {
        if (s.host) (void)hipHostFree(s.host);
        s.host = nullptr;
        HIP_TRY(hipHostMalloc((void**)&s.host, need_ints * 4, hipHostMallocDefault));
        s.host_ints = need_ints;
    }
    if (!s.done) HIP_TRY(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    if (!s.released) HIP_TRY(hipEventCreateWithFlags(&s.released, hipEventDisableTiming));
    const PlanLayout L = plan_layout(h, s, N, seq);
    const PlanArrays& p = L.p;
    HIP_TRY(hipMemsetAsync(s.i32.p, 0, L.n_clear * 4, st));          // id_flag, pair_flag, err: one piece of the arena
    const int rb = (int)((N + 255) / 256);
    hipLaunchKernelGGL(plan_rows_kernel, dim3(rb), dim3(256), 0, st, sfm, N, seq, c.pad_token_id, lam, V, p);
    launch_exclusive_scan(p.row_count, p.row_offset, N, L.scan_tmp, st);
    launch_exclusive_scan(p.id_flag, p.id_slot, (int64_t)V, L.scan_tmp, st);
    {
        // (ABI 8) a forward on a caller-provided table: a token's table slot is its id's slot in the GLOBAL distinct-id list
        PlanArrays pt = p;
        if (h->ext.id_slot) pt.id_slot = const_cast<int32_t*>(h->ext.id_slot);
        hipLaunchKernelGGL(plan_tokens_kernel, dim3(rb), dim3(256), 0, st, sfm, N, seq, c.pad_token_id, lam, V, pt);
    }
    if (L.pair_plan) {
        launch_exclusive_scan(p.pair_flag, p.pair_slot, L.PK, L.scan_tmp, st);
        hipLaunchKernelGGL(plan_pairs_kernel, dim3((unsigned)((max_tok + 255) / 256)), dim3(256), 0, st, N, p);
    }
    hipLaunchKernelGGL(plan_idlist_kernel, dim3((V + 255) / 256), dim3(256), 0, st, V, p);
    HIP_TRY(hipGetLastError());
    // the row offsets and, right behind them, the three counters (distinct ids, error word, distinct pairs) to the host: one copy
    HIP_TRY(hipMemcpyAsync(s.host, p.row_offset, ((size_t)N + 1 + 3) * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipEventRecord(s.done, st));
    s.sfm = sfm; s.n_rows = N; s.seq = seq; s.pair_plan = L.pair_plan; s.ext_id_slot = h->ext.id_slot;
    return 0;
}

template <typename T>
int do_forward(zett_hypernet* h, const int32_t* sfm, int64_t N, int seq, const void* src, int src_dtype,
               int64_t v_src, int lang_index, float* out_in, float* out_out, float* out_bias, hipStream_t st);
template <typename T>
int do_table_rows(zett_hypernet* h, const int32_t* id_list, int first, int count, const void* src, int src_dtype, void* table_out, float* stats_out, hipStream_t st);

}  // namespace

// ---------------------------------------------------------------------------------
extern "C" {

const char* zett_last_error(void) { return g_err.c_str(); }
int zett_abi_version(void) { return ZETT_ABI_VERSION; }

int zett_create(const zett_config* cfg, int device, int precision, zett_hypernet** out) {
    if (!cfg || !out) return fail(ZETT_E_INVALID, "null argument");
    if (precision != ZETT_PREC_BF16 && precision != ZETT_PREC_F32 && precision != ZETT_PREC_F16) return fail(ZETT_E_INVALID, "unknown precision %d", precision);
    if (int rc = validate_config(*cfg, precision)) return rc;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(ZETT_E_INVALID, "device %d out of range (%d visible)", device, ndev);
    ZETT_ON_DEVICE(device);
    auto* h = new zett_hypernet();
    h->cfg = *cfg;
    h->device = device;
    h->precision = precision;
    if (hipMalloc((void**)&h->range_word, 64) != hipSuccess || hipHostMalloc((void**)&h->range_host, 64, hipHostMallocDefault) != hipSuccess ||
        hipMemset(h->range_word, 0, 64) != hipSuccess) {
        if (h->range_word) (void)hipFree(h->range_word);
        if (h->range_host) (void)hipHostFree(h->range_host);
        delete h;
        return fail(ZETT_E_HIP, "allocation of the range word failed");
    }
    *out = h;
    return 0;
}

int zett_destroy(zett_hypernet* h) {
    if (!h) return 0;
    ::zett::DeviceScope _scope(h->device);
    for (auto& kv : h->w) {
        if (kv.second.f32) (void)hipFree(kv.second.f32);
        if (kv.second.lo && kv.second.lo != (void*)kv.second.f32) (void)hipFree(kv.second.lo);
    }
    for (void* p : h->owned) (void)hipFree(p);
    for (DevBuf* b : {&h->table, &h->x0, &h->yf, &h->yt, &h->big, &h->pre, &h->ctx, &h->cf, &h->ct, &h->lnstats, &h->lnparts})
        b->release();
    for (zett_hypernet::PlanSlot* psp : {&h->plan[0], &h->plan[1], &h->table_plan}) {
        zett_hypernet::PlanSlot& ps = *psp;
        ps.i32.release(); ps.u8.release();
        if (ps.host) (void)hipHostFree(ps.host);
        if (ps.done) (void)hipEventDestroy(ps.done);
        if (ps.released) (void)hipEventDestroy(ps.released);
    }
    if (h->plan_stream) (void)hipStreamDestroy(h->plan_stream);
    if (h->plan_fork) (void)hipEventDestroy(h->plan_fork);
    if (h->range_word) (void)hipFree(h->range_word);
    if (h->range_host) (void)hipHostFree(h->range_host);
    for (hipEvent_t e : h->out_ready) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->lane_ev) if (e) (void)hipEventDestroy(e);
    if (h->lane_stream) (void)hipStreamDestroy(h->lane_stream);
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    delete h;
    return 0;
}

int zett_load_weight(zett_hypernet* h, const char* name, const void* data, int dtype, const int64_t* shape, int ndim) {
    if (!h || !name || !data || !shape || ndim < 1) return fail(ZETT_E_INVALID, "null argument");
    if (h->finalized) return fail(ZETT_E_STATE, "weights are frozen after zett_finalize");
    std::map<std::string, std::vector<int64_t>> exp;
    expected_shapes(h->cfg, exp);
    const std::string n(name);
    auto it = exp.find(n);
    if (it == exp.end()) return 0;   // e.g. model.embeddings.word_embeddings.weight: never read
    std::vector<int64_t> got(shape, shape + ndim);
    if (got != it->second) {
        std::string a, b;
        for (auto v : got) a += std::to_string(v) + ",";
        for (auto v : it->second) b += std::to_string(v) + ",";
        return fail(ZETT_E_INVALID, "%s: shape [%s] does not match the config's [%s]", name, a.c_str(), b.c_str());
    }
    ZETT_ON_DEVICE(h->device);
    size_t numel = 1;
    for (auto v : got) numel *= (size_t)v;
    Tensor& t = h->w[n];
    if (t.f32) { (void)hipFree(t.f32); t.f32 = nullptr; }
    t.shape = got;
    t.numel = numel;
    HIP_TRY(hipMalloc((void**)&t.f32, numel * 4));
    if (dtype == ZETT_F32) {
        HIP_TRY(hipMemcpy(t.f32, data, numel * 4, hipMemcpyDefault));
    } else if (dtype == ZETT_F16 || dtype == ZETT_BF16) {
        void* stage = nullptr;
        HIP_TRY(hipMalloc(&stage, numel * 2));
        hipError_t e = hipMemcpy(stage, data, numel * 2, hipMemcpyDefault);
        if (e == hipSuccess) {
            const int blocks = (int)std::min<size_t>((numel + 255) / 256, 65535);
            if (dtype == ZETT_F16) hipLaunchKernelGGL(convert_to_f32_kernel<1>, dim3(blocks), dim3(256), 0, 0, stage, t.f32, numel);
            else hipLaunchKernelGGL(convert_to_f32_kernel<2>, dim3(blocks), dim3(256), 0, 0, stage, t.f32, numel);
            e = hipDeviceSynchronize();
        }
        (void)hipFree(stage);
        if (e != hipSuccess) return fail(ZETT_E_HIP, "upload of %s failed: %s", name, hipGetErrorString(e));
    } else {
        return fail(ZETT_E_INVALID, "unknown dtype %d", dtype);
    }
    return 0;
}

int zett_finalize(zett_hypernet* h) {
    if (!h) return fail(ZETT_E_INVALID, "null handle");
    if (h->finalized) return 0;
    ZETT_ON_DEVICE(h->device);
    std::map<std::string, std::vector<int64_t>> exp;
    expected_shapes(h->cfg, exp);
    for (auto& kv : exp)
        if (!h->w.count(kv.first)) return fail(ZETT_E_STATE, "missing weight: %s", kv.first.c_str());
    const zett_config& c = h->cfg;
    const size_t es = elt_size(h->precision);
    // GEMM operands in the arithmetic type
    for (auto& kv : h->w) {
        Tensor& t = kv.second;
        if (!is_gemm_weight(kv.first)) continue;
        if (h->precision == ZETT_PREC_F32) { t.lo = t.f32; continue; }
        HIP_TRY(hipMalloc(&t.lo, t.numel * 2));
        const int blocks = (int)std::min<size_t>((t.numel / 4 + 255) / 256 + 1, 65535);
        if (h->precision == ZETT_PREC_F16)
            hipLaunchKernelGGL(convert_f32_to_lo_kernel<f16_t>, dim3(blocks), dim3(256), 0, 0, t.f32, (f16_t*)t.lo, t.numel, h->range_word);
        else
            hipLaunchKernelGGL(convert_f32_to_lo_kernel<bf16_t>, dim3(blocks), dim3(256), 0, 0, t.f32, (bf16_t*)t.lo, t.numel, h->range_word);
    }
    HIP_TRY(hipDeviceSynchronize());
    // fused QKV operand per layer: rows [q | k | v]
    const size_t H = c.hidden;
    for (int l = 0; l < c.layers; ++l) {
        const std::string p = "model.encoder.layer." + std::to_string(l) + ".attention.self.";
        void* wq = nullptr; float* bq = nullptr;
        HIP_TRY(hipMalloc(&wq, 3 * H * H * es));
        HIP_TRY(hipMalloc((void**)&bq, 3 * H * 4));
        h->owned.push_back(wq); h->owned.push_back(bq);
        int k = 0;
        for (const char* q : {"query", "key", "value"}) {
            Tensor& tw = h->w[p + q + ".weight"];
            Tensor& tb = h->w[p + q + ".bias"];
            HIP_TRY(hipMemcpy((char*)wq + (size_t)k * H * H * es, tw.lo, H * H * es, hipMemcpyDeviceToDevice));
            HIP_TRY(hipMemcpy(bq + (size_t)k * H, tb.f32, H * 4, hipMemcpyDeviceToDevice));
            ++k;
        }
        h->qkv_w.push_back(wq);
        h->qkv_b.push_back(bq);
    }
    // LayerNorm fold operands (rowops.hip.h fold_weight_kernel); needs the fp32 originals, which are freed below
    if (h->precision != ZETT_PREC_F32 && c.hidden % 128 == 0) {
        auto fold = [&](const float* w32, size_t N, size_t K, const float* gamma, const float* beta, const float* bias,
                        zett_hypernet::Folded& f) -> int {
            HIP_TRY(hipMalloc(&f.w, N * K * es));
            HIP_TRY(hipMalloc((void**)&f.c, N * 4));
            HIP_TRY(hipMalloc((void**)&f.b, N * 4));
            h->owned.push_back(f.w); h->owned.push_back(f.c); h->owned.push_back(f.b);
            if (h->precision == ZETT_PREC_F16)
                hipLaunchKernelGGL(fold_weight_kernel<f16_t>, dim3((unsigned)N), dim3(256), 0, 0, w32, (int)K, gamma, beta, bias, (f16_t*)f.w, f.c, f.b, h->range_word);
            else
                hipLaunchKernelGGL(fold_weight_kernel<bf16_t>, dim3((unsigned)N), dim3(256), 0, 0, w32, (int)K, gamma, beta, bias, (bf16_t*)f.w, f.c, f.b, h->range_word);
            return 0;
        };
        h->fold_qkv.resize(c.layers); h->fold_up.resize(c.layers);
        float* tmp = nullptr;                       // fp32 [3H, H] fused QKV of one layer
        HIP_TRY(hipMalloc((void**)&tmp, 3 * H * H * 4));
        for (int l = 0; l < c.layers; ++l) {
            const std::string lp = "model.encoder.layer." + std::to_string(l) + ".";
            if (int rc = fold(h->w[lp + "intermediate.dense.weight"].f32, c.intermediate, H, h->w[lp + "attention.output.LayerNorm.weight"].f32,
                              h->w[lp + "attention.output.LayerNorm.bias"].f32, h->w[lp + "intermediate.dense.bias"].f32, h->fold_up[l])) return rc;
            if (l == 0) continue;
            const std::string pp = "model.encoder.layer." + std::to_string(l - 1) + ".output.LayerNorm.";
            int k = 0;
            for (const char* q : {"query", "key", "value"}) {
                HIP_TRY(hipMemcpy(tmp + (size_t)k * H * H, h->w[lp + "attention.self." + q + ".weight"].f32, H * H * 4, hipMemcpyDeviceToDevice));
                ++k;
            }
            if (int rc = fold(tmp, 3 * H, H, h->w[pp + "weight"].f32, h->w[pp + "bias"].f32, h->qkv_b[l], h->fold_qkv[l])) return rc;
        }
        // the output heads: Linear(LN_1e-6(.)) with the ProjectorBlock's LayerNorm folded in (not for the split single head,
        // whose two-destination epilogue is the generic one)
        if (!(c.single_head && c.separate_out)) {
            const size_t w0 = c.single_head ? c.n_in_embd : c.n_embd;
            if (int rc = fold(h->w["output_projection.1.weight"].f32, w0, H, h->w["output_projection.0.ln.weight"].f32,
                              h->w["output_projection.0.ln.bias"].f32, h->w["output_projection.1.bias"].f32, h->fold_head_in)) return rc;
            if (c.separate_out && !c.single_head)
                if (int rc = fold(h->w["output_projection_out.1.weight"].f32, c.n_embd, H, h->w["output_projection_out.0.ln.weight"].f32,
                                  h->w["output_projection_out.0.ln.bias"].f32, h->w["output_projection_out.1.bias"].f32, h->fold_head_out)) return rc;
            std::vector<float> ones(w0, 1.0f), zeros(w0, 0.0f);
            HIP_TRY(hipMalloc((void**)&h->head_one, w0 * 4));
            HIP_TRY(hipMalloc((void**)&h->head_zero, w0 * 4));
            h->owned.push_back(h->head_one); h->owned.push_back(h->head_zero);
            HIP_TRY(hipMemcpy(h->head_one, ones.data(), w0 * 4, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(h->head_zero, zeros.data(), w0 * 4, hipMemcpyHostToDevice));
        }
        HIP_TRY(hipDeviceSynchronize());
        (void)hipFree(tmp);
    }
    // range guard: a weight (or gamma-folded weight) that does not fit the operand type is an error here, not an inf later
    HIP_TRY(hipMemcpy(h->range_host, h->range_word, 4, hipMemcpyDeviceToHost));
    if (h->range_host[0] & ZETT_RANGE_WEIGHT)
        return fail(ZETT_E_RANGE, "a GEMM weight (or a LayerNorm-folded weight W*gamma) exceeds the half range (|w| > 65504 or not finite): "
                                  "this checkpoint needs ZETT_PREC_BF16 or ZETT_PREC_F32");
    // output Rescaler vectors laid out over the first head's columns
    if (c.rescale) {
        const size_t w0 = c.single_head ? c.n_in_embd : c.n_embd;
        HIP_TRY(hipMalloc((void**)&h->head_scale, w0 * 4));
        HIP_TRY(hipMalloc((void**)&h->head_shift, w0 * 4));
        h->owned.push_back(h->head_scale); h->owned.push_back(h->head_shift);
        HIP_TRY(hipMemcpy(h->head_scale, h->w["scaler.w"].f32, c.n_embd * 4, hipMemcpyDeviceToDevice));
        HIP_TRY(hipMemcpy(h->head_shift, h->w["scaler.b"].f32, c.n_embd * 4, hipMemcpyDeviceToDevice));
        if (c.single_head && c.separate_out) {
            HIP_TRY(hipMemcpy(h->head_scale + c.n_embd, h->w["out_scaler.w"].f32, c.n_embd * 4, hipMemcpyDeviceToDevice));
            HIP_TRY(hipMemcpy(h->head_shift + c.n_embd, h->w["out_scaler.b"].f32, c.n_embd * 4, hipMemcpyDeviceToDevice));
        }
    }
    // the fp32 originals of bf16 GEMM operands are no longer needed
    if (h->precision != ZETT_PREC_F32) {
        for (auto& kv : h->w) {
            Tensor& t = kv.second;
            if (is_gemm_weight(kv.first) && t.f32) { (void)hipFree(t.f32); t.f32 = nullptr; }
        }
    }
    h->finalized = true;
    return 0;
}

int zett_set_option(zett_hypernet* h, const char* key, int64_t value) {
    if (!h || !key) return fail(ZETT_E_INVALID, "null argument");
    const std::string k(key);
    if (k == "max_chunk_tokens") {
        if (value < 1024) return fail(ZETT_E_INVALID, "max_chunk_tokens must be >= 1024");
        h->max_chunk_tokens = value;
    } else if (k == "time_gemm") {
        h->time_gemm = value != 0;
    } else if (k == "cls_only_last_layer") {
        h->cls_only_last = value != 0;
    } else if (k == "pair_dedupe") {
        h->pair_dedupe = value != 0;
    } else if (k == "residual_lo") {
        if (value < 0 || value > 2) return fail(ZETT_E_INVALID, "residual_lo must be 0 (fp32 residual stream), 1 (16-bit stream in f16 mode) or 2 (in bf16 mode too)");
        h->residual_lo = (int)value;
    } else if (k == "concurrent_lanes") {
        if (value < 0 || value > 2) return fail(ZETT_E_INVALID, "concurrent_lanes must be 0 (never), 1 (auto) or 2 (whenever a call is one chunk of >= 512 rows)");
        h->concurrent_lanes = (int)value;
    } else if (k == "attention_fast") {
        h->attention_fast = value != 0;
    } else if (k == "ln_fold") {
        if (value < 0 || value > 2) return fail(ZETT_E_INVALID, "ln_fold must be 0 (off), 1 (encoder and output heads) or 2 (encoder only: A/B)");
        h->ln_fold = (int)value;
    } else if (k == "gemm_tail_split") {
        if (value < 0 || value > 3) return fail(ZETT_E_INVALID, "gemm_tail_split must be 0 (never), 1 (auto), 2 (cut every gemm4d launch in the middle) or 3 (128x256 tiles only)");
        h->gemm_tail_split = (int)value;
    } else if (k == "attention_pack") {
        h->attention_pack = value != 0;
    } else if (k == "ln_rows8") {
        h->ln_rows8 = value != 0;
    } else if (k == "table_lo") {
        h->table_lo = value != 0;
    } else if (k == "gemm_group") {
        if (value < 0 || value > 64) return fail(ZETT_E_INVALID, "gemm_group must be 0 (default: 4) .. 64");
        h->gemm_group = (int)value;
    } else if (k == "gemm_tile_order") {
        if (value < 0 || value > 1) return fail(ZETT_E_INVALID, "gemm_tile_order must be 0 or 1");
        h->gemm_tile_order = (int)value;
    } else if (k == "gemm4d_min_k") {
        if (value < 64) return fail(ZETT_E_INVALID, "gemm4d_min_k must be >= 64");
        h->gemm4d_min_k = (int)value;
    } else if (k == "range_accumulate") {
        h->range_accumulate = value != 0;
    } else if (k == "gemm_variant") {
        if (value != 0 && value != 1 && value != 2 && value != 3 && value != 7 && value != 8)
            return fail(ZETT_E_INVALID, "gemm_variant must be 0 (auto), 1 (128x128), 2 (256x256 register-staged), 3 (384x256), 7 (256x256 four-wave direct-to-LDS) or 8 (7 with the generic epilogue drain)");
        h->gemm_variant = (int)value;
    } else {
        return fail(ZETT_E_INVALID, "unknown option %s", key);
    }
    return 0;
}

int zett_workspace_bytes(const zett_hypernet* h, int64_t n_rows, int32_t seq, int64_t* out_bytes) {
    if (!h || !out_bytes) return fail(ZETT_E_INVALID, "null argument");
    if (n_rows < 0 || seq < 1) return fail(ZETT_E_INVALID, "bad surface-form shape [%lld, %d]", (long long)n_rows, seq);
    const zett_config& c = h->cfg;
    const int64_t V = (int64_t)c.original_vocab_size + c.n_extra;
    const int64_t max_tok = n_rows * (int64_t)(seq + (c.embed_lang ? 1 : 0));
    // worst case of the plan: no pad position, every position a different source id
    const WorkspaceSizes w = workspace_sizes(c, elt_size(h->precision), seq, max_tok, std::min<int64_t>(V, max_tok), h->max_chunk_tokens);
    const size_t parts = c.hidden % 128 == 0 ? (size_t)(c.hidden / 128) * ((size_t)w.chunk_tokens + 768) * 8 : 0;      // LayerNorm-fold partials
    *out_bytes = (int64_t)(w.total() + parts + plan_i32_bytes(c, n_rows, seq) + (size_t)n_rows + (size_t)max_tok);
    return 0;
}

int zett_get_stats(const zett_hypernet* h, zett_stats* out) {
    if (!h || !out) return fail(ZETT_E_INVALID, "null argument");
    *out = h->stats;
    return 0;
}

int zett_stream_wait_output(zett_hypernet* h, int which, void* stream) {
    if (!h) return fail(ZETT_E_INVALID, "null handle");
    if (which != ZETT_OUT_IN && which != ZETT_OUT_BIAS) return fail(ZETT_E_INVALID, "which must be ZETT_OUT_IN or ZETT_OUT_BIAS");
    if (!h->out_recorded) return fail(ZETT_E_STATE, "no forward has run on this handle");
    ZETT_ON_DEVICE(h->device);
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, h->out_ready[which], 0));
    return 0;
}

int zett_get_gemm_log(const zett_hypernet* h, zett_gemm_record* out, int64_t capacity, int64_t* count) {
    if (!h || !count || capacity < 0 || (capacity > 0 && !out)) return fail(ZETT_E_INVALID, "null argument");
    *count = (int64_t)h->gemm_log.size();
    for (int64_t i = 0; i < capacity && i < *count; ++i) out[i] = h->gemm_log[(size_t)i];
    return 0;
}

int zett_check_range(zett_hypernet* h, void* stream, int32_t* flags) {
    if (!h) return fail(ZETT_E_INVALID, "null handle");
    ZETT_ON_DEVICE(h->device);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(h->range_host, h->range_word, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const int32_t w = h->range_host[0];
    if (h->range_accumulate && w) HIP_TRY(hipMemsetAsync(h->range_word, 0, 4, st));
    if (flags) *flags = w;
    if (!w) return 0;
    return fail(ZETT_E_RANGE, "the last forward left the range of its arithmetic:%s%s%s (precision %s)",
                (w & ZETT_RANGE_SOURCE) ? " in_scaler(source_embeddings) beyond the half range;" : "",
                (w & ZETT_RANGE_ACTIVATION) ? " a 16-bit activation (Q/K/V, FFN intermediate or the operand copy of the residual sum) beyond the half range;" : "",
                (w & ZETT_RANGE_OUTPUT) ? " non-finite predicted embeddings;" : "",
                h->precision == ZETT_PREC_F16 ? "f16: re-run with ZETT_PREC_BF16" : h->precision == ZETT_PREC_BF16 ? "bf16" : "f32");
}

int zett_partition_workspace_bytes(int64_t n_rows, int32_t n_ids, int64_t* out_bytes) {
    if (!out_bytes || n_rows < 0 || n_ids < 1) return fail(ZETT_E_INVALID, "bad argument");
    *out_bytes = (int64_t)partition_workspace_bytes(n_rows, n_ids);
    return 0;
}

int zett_partition_rows(const int32_t* surface_forms, int64_t n_rows, int32_t seq, int32_t pad_id, int32_t n_ids, int32_t world,
                        const int32_t* caps, int32_t* perm_out, void* workspace, int64_t workspace_bytes, int32_t device, void* stream) {
    if (n_rows < 0 || seq < 1 || n_ids < 1) return fail(ZETT_E_INVALID, "bad surface-form shape [%lld, %d] / id range %d", (long long)n_rows, seq, n_ids);
    if (world < 1 || world > PART_MAX_RANKS) return fail(ZETT_E_INVALID, "zett_partition_rows handles 1..%d ranks (one node), got %d", PART_MAX_RANKS, world);
    if (n_rows == 0) return 0;
    if (!surface_forms || !caps || !perm_out || !workspace) return fail(ZETT_E_INVALID, "null argument");
    if (n_rows >= (int64_t)0x7fffffff) return fail(ZETT_E_INVALID, "too many rows for one call");
    int64_t total = 0;
    for (int r = 0; r < world; ++r) { if (caps[r] < 0) return fail(ZETT_E_INVALID, "negative capacity"); total += caps[r]; }
    if (total != n_rows) return fail(ZETT_E_INVALID, "the ranks' capacities sum to %lld, the matrix has %lld rows", (long long)total, (long long)n_rows);
    if ((size_t)workspace_bytes < partition_workspace_bytes(n_rows, n_ids)) return fail(ZETT_E_INVALID, "workspace too small (zett_partition_workspace_bytes)");
    ZETT_ON_DEVICE(device);
    hipStream_t st = (hipStream_t)stream;
    const size_t have_bytes = ((size_t)(n_ids + 3) / 4) * 4 + 64;
    uint32_t* have = (uint32_t*)workspace;
    PartCaps caps_arg = {};
    for (int r = 0; r < world; ++r) caps_arg.v[r] = caps[r];
    HIP_TRY(hipMemsetAsync(have, 0, have_bytes, st));
    // the rank bytes live in LDS when the id range fits (a byte per id beside ~1 KiB of counters), else in the workspace
    const size_t lds_have = ((size_t)(n_ids + 3) / 4) * 4;
    if (lds_have <= 150 * 1024) {
        static bool attr_set[64] = {};          // (per device: the attribute belongs to the device's copy of the kernel)
        if (device >= 64 || !attr_set[device]) {
            HIP_TRY(hipFuncSetAttribute((const void*)partition_rows_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            if (device < 64) attr_set[device] = true;
        }
        hipLaunchKernelGGL(partition_rows_kernel<true>, dim3(1), dim3(PART_THREADS), lds_have, st, surface_forms, n_rows, (int)seq, (int)pad_id, (int)n_ids,
                           (int)world, caps_arg, have, perm_out);
    } else {
        hipLaunchKernelGGL(partition_rows_kernel<false>, dim3(1), dim3(PART_THREADS), 0, st, surface_forms, n_rows, (int)seq, (int)pad_id, (int)n_ids,
                           (int)world, caps_arg, have, perm_out);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int zett_scatter_rows(const void* src, void* dst, const int64_t* order, int64_t n_rows, int64_t row_bytes, int32_t device, void* stream) {
    if (n_rows < 0 || row_bytes < 1) return fail(ZETT_E_INVALID, "bad shape");
    if (n_rows == 0) return 0;
    if (!src || !dst || !order) return fail(ZETT_E_INVALID, "null argument");
    if (row_bytes != 4 && row_bytes % 16 != 0) return fail(ZETT_E_INVALID, "row_bytes must be 4 or a multiple of 16, got %lld", (long long)row_bytes);
    if (((uintptr_t)src | (uintptr_t)dst) % (row_bytes == 4 ? 4 : 16)) return fail(ZETT_E_INVALID, "misaligned buffer");
    ZETT_ON_DEVICE(device);
    hipStream_t st = (hipStream_t)stream;
    if (row_bytes == 4)
        hipLaunchKernelGGL(scatter_words_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, st, (const uint32_t*)src, (uint32_t*)dst, order, n_rows);
    else
        hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)std::min<int64_t>(n_rows, 65535 * 16)), dim3(256), 0, st, (const unsigned char*)src, (unsigned char*)dst,
                           order, n_rows, row_bytes);
    HIP_TRY(hipGetLastError());
    return 0;
}

int zett_forward_prepare(zett_hypernet* h, const int32_t* surface_forms, int64_t n_rows, int32_t seq, void* input_stream) {
    if (!h) return fail(ZETT_E_INVALID, "null handle");
    if (!h->finalized) return fail(ZETT_E_STATE, "zett_finalize has not been called");
    const zett_config& c = h->cfg;
    if (n_rows < 0 || seq < 1) return fail(ZETT_E_INVALID, "bad surface-form shape [%lld, %d]", (long long)n_rows, seq);
    if (n_rows == 0) return 0;
    if (!surface_forms) return fail(ZETT_E_INVALID, "null tensor argument");
    if (n_rows * (int64_t)(seq + 1) >= (int64_t)0x7fffffff) return fail(ZETT_E_INVALID, "too many positions for one call");
    (void)c;
    ZETT_ON_DEVICE(h->device);
    if (!h->plan_stream) {      // highest priority: the plan's small workgroups must get CUs while a forward's GEMM tiles own the chip
        int least = 0, greatest = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_TRY(hipStreamCreateWithPriority(&h->plan_stream, hipStreamNonBlocking, greatest));
    }
    if (!h->plan_fork) HIP_TRY(hipEventCreateWithFlags(&h->plan_fork, hipEventDisableTiming));
    zett_hypernet::PlanSlot& ps = h->plan[h->plan_cur ^ 1];
    if (ps.pending) HIP_TRY(hipEventSynchronize(ps.done));           // replaces a prepared plan nobody took
    // behind the surface forms (whatever input_stream holds now) and behind the forward that last read this slot
    HIP_TRY(hipEventRecord(h->plan_fork, (hipStream_t)input_stream));
    HIP_TRY(hipStreamWaitEvent(h->plan_stream, h->plan_fork, 0));
    if (ps.released) HIP_TRY(hipStreamWaitEvent(h->plan_stream, ps.released, 0));
    if (int rc = enqueue_plan(h, ps, surface_forms, n_rows, seq, h->plan_stream)) return rc;
    ps.pending = true;
    return 0;
}

int zett_forward(zett_hypernet* h, const int32_t* surface_forms, int64_t n_rows, int32_t seq,
                 const void* source_embeddings, int src_dtype, int64_t v_src, int32_t lang_index,
                 float* out_in, float* out_out, float* out_bias, void* stream) {
    if (!h) return fail(ZETT_E_INVALID, "null handle");
    if (!h->finalized) return fail(ZETT_E_STATE, "zett_finalize has not been called");
    const zett_config& c = h->cfg;
    if (n_rows < 0 || seq < 1) return fail(ZETT_E_INVALID, "bad surface-form shape [%lld, %d]", (long long)n_rows, seq);
    if (seq + (c.embed_lang ? 1 : 0) > c.max_positions) return fail(ZETT_E_INDEX, "sequence %d exceeds position_embeddings (%d rows)", seq, c.max_positions);
    if (n_rows == 0) {
        h->stats = zett_stats{};
        ZETT_ON_DEVICE(h->device);
        if (!h->range_accumulate) HIP_TRY(hipMemsetAsync(h->range_word, 0, 4, (hipStream_t)stream));
        for (hipEvent_t& e : h->out_ready) {
            if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(e, (hipStream_t)stream));
        }
        h->out_recorded = true;
        return 0;
    }
    if (!surface_forms || !source_embeddings || !out_in || !out_bias) return fail(ZETT_E_INVALID, "null tensor argument");
    const bool has_out = c.separate_out;
    if (has_out && !out_out) return fail(ZETT_E_INVALID, "out_out is required when separate_out_embeddings is set");
    if (src_dtype < ZETT_F32 || src_dtype > ZETT_BF16) return fail(ZETT_E_INVALID, "unknown source dtype %d", src_dtype);
    if (v_src < c.original_vocab_size) return fail(ZETT_E_INDEX, "source_embeddings has %lld rows, config.original_vocab_size is %d", (long long)v_src, c.original_vocab_size);
    if (c.embed_lang && (lang_index < 0 || lang_index >= c.n_langs)) return fail(ZETT_E_INDEX, "lang_index %d outside [0,%d)", lang_index, c.n_langs);
    if (n_rows * (int64_t)(seq + 1) >= (int64_t)0x7fffffff) return fail(ZETT_E_INVALID, "too many positions for one call");
    ZETT_ON_DEVICE(h->device);
    hipStream_t st = (hipStream_t)stream;
    if (h->precision == ZETT_PREC_F16)
        return do_forward<f16_t>(h, surface_forms, n_rows, seq, source_embeddings, src_dtype, v_src, lang_index, out_in, out_out, out_bias, st);
    if (h->precision == ZETT_PREC_BF16)
        return do_forward<bf16_t>(h, surface_forms, n_rows, seq, source_embeddings, src_dtype, v_src, lang_index, out_in, out_out, out_bias, st);
    return do_forward<float>(h, surface_forms, n_rows, seq, source_embeddings, src_dtype, v_src, lang_index, out_in, out_out, out_bias, st);
}

// ---- (ABI 8) the hoisted table shared between ranks: SURVEY 8e's optional second exchange ---------------------------------
int zett_table_plan(zett_hypernet* h, const int32_t* surface_forms, int64_t n_rows, int32_t seq, int32_t* id_slot_out, int32_t* id_list_out,
                    int64_t* n_ids_out, void* stream) {
    if (!h) return fail(ZETT_E_INVALID, "null handle");
    if (!h->finalized) return fail(ZETT_E_STATE, "zett_finalize has not been called");
    if (!surface_forms || !id_slot_out || !id_list_out || !n_ids_out) return fail(ZETT_E_INVALID, "null argument");
    if (n_rows < 1 || seq < 1) return fail(ZETT_E_INVALID, "bad surface-form shape [%lld, %d]", (long long)n_rows, seq);
    if (n_rows * (int64_t)(seq + 1) >= (int64_t)0x7fffffff) return fail(ZETT_E_INVALID, "too many positions for one call");
    const zett_config& c = h->cfg;
    const int V = c.original_vocab_size + c.n_extra;
    ZETT_ON_DEVICE(h->device);
    hipStream_t st = (hipStream_t)stream;
    zett_hypernet::PlanSlot& ps = h->table_plan;
    const zett_hypernet::ExtTable saved = h->ext;
    h->ext = zett_hypernet::ExtTable{};          // (this plan numbers the ids itself)
    const int rc = enqueue_plan(h, ps, surface_forms, n_rows, seq, st);
    h->ext = saved;
    if (rc) return rc;
    HIP_TRY(hipEventSynchronize(ps.done));
    const int32_t* hoff = ps.host;
    if (hoff[n_rows + 2] != 0)
        return fail(ZETT_E_INDEX, "surface-form row %d holds an id outside [0, %d) (original_vocab_size %d + %d fallback rows)",
                    hoff[n_rows + 2] - 1, V, c.original_vocab_size, c.n_extra);
    const int D = hoff[n_rows + 1];
    const PlanLayout L = plan_layout(h, ps, n_rows, seq);
    HIP_TRY(hipMemcpyAsync(id_slot_out, L.p.id_slot, ((size_t)V + 1) * 4, hipMemcpyDeviceToDevice, st));
    if (D > 0) HIP_TRY(hipMemcpyAsync(id_list_out, L.p.id_list, (size_t)D * 4, hipMemcpyDeviceToDevice, st));
    *n_ids_out = D;
    return 0;
}

int zett_table_rows(zett_hypernet* h, const int32_t* id_list, int64_t first, int64_t count, const void* source_embeddings, int src_dtype,
                    int64_t v_src, void* table_out, float* stats_out, void* stream) {
    if (!h) return fail(ZETT_E_INVALID, "null handle");
    if (!h->finalized) return fail(ZETT_E_STATE, "zett_finalize has not been called");
    if (first < 0 || count < 0 || first + count >= (int64_t)0x7fffffff) return fail(ZETT_E_INVALID, "bad table range [%lld, +%lld)", (long long)first, (long long)count);
    if (count == 0) return 0;
    if (!id_list || !source_embeddings || !table_out || !stats_out) return fail(ZETT_E_INVALID, "null tensor argument");
    if (src_dtype < ZETT_F32 || src_dtype > ZETT_BF16) return fail(ZETT_E_INVALID, "unknown source dtype %d", src_dtype);
    if (v_src < h->cfg.original_vocab_size) return fail(ZETT_E_INDEX, "source_embeddings has %lld rows, config.original_vocab_size is %d", (long long)v_src, h->cfg.original_vocab_size);
    ZETT_ON_DEVICE(h->device);
    hipStream_t st = (hipStream_t)stream;
    if (h->precision == ZETT_PREC_F16) return do_table_rows<f16_t>(h, id_list, (int)first, (int)count, source_embeddings, src_dtype, table_out, stats_out, st);
    if (h->precision == ZETT_PREC_BF16) return do_table_rows<bf16_t>(h, id_list, (int)first, (int)count, source_embeddings, src_dtype, table_out, stats_out, st);
    return fail(ZETT_E_INVALID, "zett_table_rows: the folded 16-bit table exists in the 16-bit modes only");
}

int zett_forward_table(zett_hypernet* h, const int32_t* surface_forms, int64_t n_rows, int32_t seq, const void* table, const float* table_stats,
                       const int32_t* id_slot, int32_t lang_index, float* out_in, float* out_out, float* out_bias, void* stream) {
    if (!h) return fail(ZETT_E_INVALID, "null handle");
    if (!h->finalized) return fail(ZETT_E_STATE, "zett_finalize has not been called");
    const zett_config& c = h->cfg;
    if (n_rows < 1 || seq < 1) return fail(ZETT_E_INVALID, "bad surface-form shape [%lld, %d]", (long long)n_rows, seq);
    if (seq + (c.embed_lang ? 1 : 0) > c.max_positions) return fail(ZETT_E_INDEX, "sequence %d exceeds position_embeddings (%d rows)", seq, c.max_positions);
    if (!surface_forms || !table || !table_stats || !id_slot || !out_in || !out_bias) return fail(ZETT_E_INVALID, "null tensor argument");
    if (c.separate_out && !out_out) return fail(ZETT_E_INVALID, "out_out is required when separate_out_embeddings is set");
    if (c.embed_lang && (lang_index < 0 || lang_index >= c.n_langs)) return fail(ZETT_E_INDEX, "lang_index %d outside [0,%d)", lang_index, c.n_langs);
    if (n_rows * (int64_t)(seq + 1) >= (int64_t)0x7fffffff) return fail(ZETT_E_INVALID, "too many positions for one call");
    if (h->precision == ZETT_PREC_F32) return fail(ZETT_E_INVALID, "zett_forward_table: the folded 16-bit table exists in the 16-bit modes only");
    ZETT_ON_DEVICE(h->device);
    hipStream_t st = (hipStream_t)stream;
    h->ext.table = table; h->ext.stats = table_stats; h->ext.id_slot = id_slot;
    int rc;
    if (h->precision == ZETT_PREC_F16)
        rc = do_forward<f16_t>(h, surface_forms, n_rows, seq, nullptr, ZETT_F32, 0, lang_index, out_in, out_out, out_bias, st);
    else
        rc = do_forward<bf16_t>(h, surface_forms, n_rows, seq, nullptr, ZETT_F32, 0, lang_index, out_in, out_out, out_bias, st);
    h->ext = zett_hypernet::ExtTable{};
    return rc;
}

}  // extern "C"

// ---------------------------------------------------------------------------------
namespace {

template <typename T>
struct Runner {
    zett_hypernet* h;
    hipStream_t st;
    int rc = 0;

    const Tensor& W(const std::string& n) { return h->w.at(n); }
    const T* Wlo(const std::string& n) { return (const T*)h->w.at(n).lo; }
    const float* Wf(const std::string& n) { return h->w.at(n).f32; }

    GemmEpilogue<T> epi() {
        GemmEpilogue<T> e{};
        e.split_col = 0x7fffffff;
        e.range_flag = h->range_word;
        return e;
    }

    long a_rows_readable = 0;   // rows every A operand buffer can be read for (workspace slack)

    void gemm(const T* A, int lda, const T* Wp, int ldw, int M, int N, int K, const GemmEpilogue<T>& e) {
        if (rc || M <= 0) return;
        GemmArgs<T> g{A, lda, Wp, ldw, M, N, K, e};
        g.tile_order = h->gemm_tile_order;
        g.group = h->gemm_group;
        g.row0 = h->gemm_tail_split == 1 ? -1 : h->gemm_tail_split == 2 ? ((M / 2 + 255) / 256) * 256 : h->gemm_tail_split == 3 ? -2 : 0;
        const double fl = 2.0 * (double)M * (double)N * (double)K;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (h->time_gemm) {
            while (h->ev.size() < h->ev_used + 2) {
                hipEvent_t ev;
                if (hipEventCreate(&ev) != hipSuccess) { rc = fail(ZETT_E_HIP, "hipEventCreate failed"); return; }
                h->ev.push_back(ev);
            }
            e0 = h->ev[h->ev_used++];
            e1 = h->ev[h->ev_used++];
            h->ev_flops.push_back(fl);
            h->ev_shape.push_back({M, N, K, 0});
            (void)hipEventRecord(e0, st);
        }
        // Tile choice.  Small problems: 128x128.  Otherwise the 256x256 register-staged eight-wave
        // kernel (gemm8r), unless the 384x256 LDS-DMA tile needs fewer rounds over the 256 CUs (it
        // runs ~1.7x as long per tile): wave quantisation decides, e.g. M = 5 111 at N = 4096.
        // The 384-row kernel has no registers to spare for a residual or scale/shift epilogue
        // (168 per wave: it would spill to scratch, and no kernel with scratch is ever launched)
        // and does not clamp rows, so A must have
        // `a_rows_readable` >= tiles*384 rows (every A operand here is a workspace buffer with
        // that slack) and N must be a multiple of 256.  (The kernels that led to gemm8r and gemm4d
        // -- four-wave register-staged, eight-wave LDS-DMA, gemm8r on 16x16x32 MFMAs -- live in
        // tools/experiments/ with tools/gemm_bench on the `experiments` branch.)
        constexpr bool is_f32 = std::is_same<T, float>::value;
        int variant = h->gemm_variant;
        if (variant == 0) {
            variant = (M > 128 && N > 128) ? 2 : 1;
            if (variant != 1 && N % 256 == 0 && !e.scale && !e.residual) {
                const long t256 = (long)((M + 255) / 256) * (N / 256), t384 = (long)((M + 383) / 384) * (N / 256);
                const double c256 = (double)((t256 + 255) / 256), c384 = 1.7 * (double)((t384 + 255) / 256);
                if (c384 < c256) variant = 3;
            }
        }
        if (variant == 3 && (N % 256 != 0 || (long)((M + 383) / 384) * 384 > a_rows_readable || e.scale || e.shift || e.residual || e.range_final)) variant = 2;
        // 16-bit operands, K >= 2048: the four-wave direct-to-LDS tile on 16x16x32 MFMAs (4-8 % ahead of the
        // register-staged eight-wave kernels on the launches of the benchmark step; identical bits).
        if (h->gemm_variant == 0 && variant == 2 && !is_f32 && K >= h->gemm4d_min_k) variant = 7;
        if ((variant == 7 || variant == 8) && is_f32) variant = 2;
        // the large tiles drain eight columns per lane with 16-byte accesses
        const bool wide_ok = N % 8 == 0 && (!e.out_lo || e.ld_lo % 8 == 0) && e.ld_f32 % 4 == 0 && (!e.residual || e.ld_res % 4 == 0) &&
                             (e.split_col >= N || e.split_col % 8 == 0);
        if (variant != 1 && !wide_ok) variant = 1;
        if (e.residual && (e.scale || e.shift)) variant = 1;      // the large tiles compile their residual epilogues without the Rescaler
        if (e.stats_part || e.fold_stats) variant = 7;       // LayerNorm-fold launches exist in gemm4d only (any M)
        if (h->time_gemm && !h->ev_shape.empty()) h->ev_shape.back()[3] = variant;
        {
            zett_gemm_record r{};
            r.m = M; r.n = N; r.k = K; r.variant = variant;
            r.epilogue = (e.out_lo ? 1 : 0) | ((e.out_f32 || e.out_f32_b) ? 2 : 0) | (e.residual ? 4 : 0) | ((e.scale || e.shift) ? 8 : 0) |
                         (e.stats_part ? 16 : 0) | (e.fold_stats ? 32 : 0) | (e.residual_lo ? 64 : 0) | (e.act << 8);
            r.flops = fl;
            const double mn = (double)M * (double)N;
            r.bytes = ((double)M + (double)N) * (double)K * sizeof(T) + (e.out_lo ? mn * sizeof(T) : 0.0) + ((e.out_f32 || e.out_f32_b) ? mn * 4.0 : 0.0) +
                      (e.residual ? mn * 4.0 : 0.0) + (e.residual_lo ? mn * sizeof(T) : 0.0) + (e.stats_part ? (double)M * (N / 128) * 8.0 : 0.0) + ((e.fold_stats || e.res_stats) ? (double)M * 8.0 : 0.0);
            h->gemm_log.push_back(r);
        }
        const hipError_t err = launch_gemm_variant(variant, g, st);      // gemm_launch.hip.h: the tile kernels live in their own translation units
        if (h->time_gemm) (void)hipEventRecord(e1, st);
        if (err != hipSuccess) { rc = fail(ZETT_E_HIP, "gemm launch failed: %s", hipGetErrorString(err)); return; }
        h->stats.executed_flops += fl;
        h->stats.gemm_launches += 1;
    }

    void check(const char* what) {
        if (rc) return;
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) rc = fail(ZETT_E_HIP, "%s launch failed: %s", what, hipGetErrorString(e));
    }

    void layernorm(const float* in, int rows, const float* gamma, const float* beta, float eps, float* of, T* ol, float* stats = nullptr,
                   LnReadout readout = LnReadout{}, const T* in_lo = nullptr) {
        if (rc || rows <= 0) return;
        readout.in_lo = in_lo;          // (16-bit residual stream: the rows are read from the 16-bit copy of the sum; READOUT instantiations only)
        const int H = h->cfg.hidden;
        // (r6) 16-bit modes, narrow rows: eight columns per lane, 32 lanes per row up to H = 1024 (two rows per wave), 64 up to 2048
        if constexpr (sizeof(T) == 2) {
            const int tpr8 = h->ln_rows8 ? ln_rows8_tpr(H) : 0;
            if (tpr8) {
                const dim3 grid8((rows + 256 / tpr8 - 1) / (256 / tpr8));
#define ZETT_LN8_LAUNCH(TPR, RO) hipLaunchKernelGGL((layernorm_rows8_kernel<T, false, TPR, RO>), grid8, dim3(256), 0, st, in, H, rows, H, gamma, beta, eps, of, ol, stats, (float*)nullptr, LnEmbed{}, 0, readout)
                if (readout.out_bias) { if (tpr8 == 32) ZETT_LN8_LAUNCH(32, true); else ZETT_LN8_LAUNCH(64, true); }
                else { if (tpr8 == 32) ZETT_LN8_LAUNCH(32, false); else ZETT_LN8_LAUNCH(64, false); }
#undef ZETT_LN8_LAUNCH
                check("layernorm");
                return;
            }
        }
        const dim3 grid = H <= 2048 ? dim3((rows + 3) / 4) : dim3(rows);      // H <= 2048: a wave per row, four rows per workgroup
#define ZETT_LN_LAUNCH(TPR, RO) hipLaunchKernelGGL((layernorm_rows_kernel<T, false, TPR, RO>), grid, dim3(256), 0, st, in, H, rows, H, gamma, beta, eps, of, ol, stats, (float*)nullptr, LnEmbed{}, 0, readout)
        if (readout.out_bias) { if (H <= 2048) ZETT_LN_LAUNCH(64, true); else ZETT_LN_LAUNCH(256, true); }
        else { if (H <= 2048) ZETT_LN_LAUNCH(64, false); else ZETT_LN_LAUNCH(256, false); }
#undef ZETT_LN_LAUNCH
        check("layernorm");
    }

    // LayerNorm fold: (mean, rstd) per row from the partials the producer GEMM wrote
    void ln_stats(const float2* parts, int ld_part, int rows, float eps, float* stats) {
        if (rc || rows <= 0) return;
        hipLaunchKernelGGL(ln_stats_kernel, dim3((rows + 63) / 64), dim3(64), 0, st, parts, h->cfg.hidden / 128, ld_part, rows,
                           h->cfg.hidden, eps, stats);
        check("ln_stats");
    }

    // ProjectorBlock (modeling_hypernet.py:22-40) on rows already projected to H:
    //   out = LN_1e-6( gelu_t(W2·gelu_t(W1·x + b1) + b2) + x )
    // fold_parts != null (output heads, LayerNorm fold): the block's LayerNorm is not launched — dense2 writes the 16-bit copy of
    // the pre-LayerNorm sum to `ol` and partial row statistics, fold_stats receives (mean, rstd), and the caller's next GEMM
    // runs on the gamma-folded weight (no fp32 sum is written: nothing else reads it)
    void projector(const std::string& p, const T* x_lo, const float* x_f32, int rows, T* big, float* pre, float* of, T* ol,
                   float2* fold_parts = nullptr, int ld_part = 0, float* fold_stats = nullptr) {
        const zett_config& c = h->cfg;
        GemmEpilogue<T> e1 = epi();
        e1.bias = Wf(p + "dense1.bias"); e1.act = ACT_GELU_TANH; e1.out_lo = big; e1.ld_lo = c.intermediate;
        gemm(x_lo, c.hidden, Wlo(p + "dense1.weight"), c.hidden, rows, c.intermediate, c.hidden, e1);
        GemmEpilogue<T> e2 = epi();
        e2.bias = Wf(p + "dense2.bias"); e2.act = ACT_GELU_TANH; e2.residual = x_f32; e2.ld_res = c.hidden;
        e2.ld_f32 = c.hidden;
        if (fold_parts) { e2.out_lo = ol; e2.ld_lo = c.hidden; e2.stats_part = fold_parts; e2.ld_part = ld_part; }
        else e2.out_f32 = pre;
        gemm(big, c.intermediate, Wlo(p + "dense2.weight"), c.intermediate, rows, c.hidden, c.intermediate, e2);
        if (fold_parts) ln_stats(fold_parts, ld_part, rows, c.ln_eps_projector, fold_stats);
        else layernorm(pre, rows, Wf(p + "ln.weight"), Wf(p + "ln.bias"), c.ln_eps_projector, of, ol);
    }
};

template <typename T, int SD>
void launch_gather(hipStream_t st, const int32_t* id_list, int s0, int m, const void* src, const zett_config& c,
                   const float* fallback, const float* sw, const float* sb, T* out, int32_t* range_flag) {
    hipLaunchKernelGGL((gather_src_kernel<T, SD>), dim3(m), dim3(256), 0, st, id_list, s0, m, src, c.n_in_embd,
                       c.original_vocab_size, fallback, sw, sb, out, range_flag);
}

// The hoisted table (A2-A4): input_projection once per distinct source id — rows id_list[first .. first + count) into table rows
// [first, first + count) (the fp32 table `tbl32`, or — folded — the 16-bit pre-LayerNorm sums `tbl16` with (mean, rstd) in
// `tblst`), in chunks of MC rows through the caller's workspace.  Errors land in R.rc.
template <typename T>
void table_phase(Runner<T>& R, const int32_t* id_list, int first, int count, const void* src, int src_dtype, int64_t MC, size_t MCS,
                 T* X0, float* Zf, T* Zt, T* BIG, float* PRE, float2* PARTS, bool table_lo, T* tbl16, float* tblst, float* tbl32) {
    zett_hypernet* h = R.h;
    const zett_config& c = h->cfg;
    const int H = c.hidden, EIN = c.n_in_embd;
    hipStream_t st = R.st;
    const float* in_w = c.rescale ? R.Wf("in_scaler.w") : nullptr;
    const float* in_b = c.rescale ? R.Wf("in_scaler.b") : nullptr;
    for (int s0 = first; s0 < first + count && !R.rc; s0 += (int)MC) {
        const int m = (int)std::min<int64_t>(MC, first + count - s0);
        const float* fb = R.Wf("fallback_embeddings.weight");
        if (src_dtype == ZETT_F32) launch_gather<T, 0>(st, id_list, s0, m, src, c, fb, in_w, in_b, X0, h->range_word);
        else if (src_dtype == ZETT_F16) launch_gather<T, 1>(st, id_list, s0, m, src, c, fb, in_w, in_b, X0, h->range_word);
        else launch_gather<T, 2>(st, id_list, s0, m, src, c, fb, in_w, in_b, X0, h->range_word);
        R.check("gather_src");
        GemmEpilogue<T> e0 = R.epi();
        e0.bias = R.Wf("input_projection.0.bias"); e0.out_f32 = Zf; e0.ld_f32 = H; e0.out_lo = Zt; e0.ld_lo = H;
        R.gemm(X0, EIN, R.Wlo("input_projection.0.weight"), EIN, m, H, EIN, e0);
        if (table_lo) R.projector("input_projection.1.", Zt, Zf, m, BIG, PRE, nullptr, tbl16 + (size_t)s0 * H, PARTS, (int)MCS, tblst + 2 * (size_t)s0);
        else R.projector("input_projection.1.", Zt, Zf, m, BIG, PRE, tbl32 + (size_t)s0 * H, nullptr);
    }
}

// true when this handle's forward keeps the hoisted table folded (16-bit pre-LayerNorm sums + statistics): the predicate of do_forward
template <typename T>
bool folded_table_mode(const zett_hypernet* h) {
    const zett_config& c = h->cfg;
    const int H = c.hidden;
    const bool fold = h->ln_fold && !std::is_same<T, float>::value && h->gemm_variant == 0 && H % 128 == 0 && H >= 512 && (int)h->fold_up.size() == c.layers;
    const bool lo_stream = fold && c.layers >= 1 && sizeof(T) == 2 && (h->residual_lo == 2 || (h->residual_lo == 1 && std::is_same<T, f16_t>::value));
    return lo_stream && h->table_lo != 0;
}

// (ABI 8) zett_table_rows: rows [first, first + count) of the folded table of the distinct-id list `id_list` into the caller's buffers
template <typename T>
int do_table_rows(zett_hypernet* h, const int32_t* id_list, int first, int count, const void* src, int src_dtype, void* table_out, float* stats_out, hipStream_t st) {
    const zett_config& c = h->cfg;
    if (!folded_table_mode<T>(h))
        return fail(ZETT_E_INVALID, "zett_table_rows needs the folded 16-bit table: f16 arithmetic with the LayerNorm fold, the 16-bit residual stream and table_lo on");
    const WorkspaceSizes ws = workspace_sizes(c, sizeof(T), 1, count, count, h->max_chunk_tokens);
    const int64_t MC = ws.chunk_tokens;
    const size_t MCS = (size_t)MC + 768;
    if (int rc = h->x0.reserve(ws.x0)) return rc;
    if (int rc = h->yf.reserve(ws.f32_rows)) return rc;
    if (int rc = h->yt.reserve(ws.lo_rows)) return rc;
    if (int rc = h->big.reserve(ws.big)) return rc;
    if (int rc = h->pre.reserve(ws.f32_rows)) return rc;
    if (int rc = h->lnparts.reserve((size_t)(c.hidden / 128) * MCS * sizeof(float2))) return rc;
    Runner<T> R{h, st};
    R.a_rows_readable = (long)MCS;
    // (per-launch events and the launch log belong to a forward: zett_forward resets both when it starts)
    struct NoTiming { zett_hypernet* h; int saved; ~NoTiming() { h->time_gemm = saved; } } no_timing{h, h->time_gemm};
    h->time_gemm = 0;
    table_phase<T>(R, id_list, first, count, src, src_dtype, MC, MCS, h->x0.as<T>(), h->yf.as<float>(), h->yt.as<T>(), h->big.as<T>(), h->pre.as<float>(),
                   h->lnparts.as<float2>(), true, (T*)table_out, stats_out, nullptr);
    return R.rc;
}

template <typename T>
int do_forward(zett_hypernet* h, const int32_t* sfm, int64_t N, int seq, const void* src, int src_dtype,
               int64_t v_src, int lang_index, float* out_in, float* out_out, float* out_bias, hipStream_t st) {
    (void)v_src;
    const zett_config& c = h->cfg;
    const int H = c.hidden, I = c.intermediate, E = c.n_embd, EIN = c.n_in_embd;
    const int lam = c.embed_lang ? 1 : 0;
    const int V = c.original_vocab_size + c.n_extra;
    const int64_t max_tok = N * (int64_t)(seq + lam);
    h->stats = zett_stats{};
    h->stats.rows = N;
    h->ev_used = 0;
    h->ev_flops.clear();
    h->ev_shape.clear();
    h->gemm_log.clear();

    // ---- plan ---------------------------------------------------------------------
    // The slot the previous forward did not use.  Prepared for exactly this call (zett_forward_prepare): its plan ran on the
    // plan stream, the host waits for THAT (not for earlier work on st) and st is ordered behind it.  Otherwise the plan is
    // made here, on st, and the host waits for it — and so for whatever was enqueued on st before.
    zett_hypernet::PlanSlot& ps = h->plan[h->plan_cur ^ 1];
    if (ps.pending && ps.sfm == sfm && ps.n_rows == N && ps.seq == seq && ps.pair_plan == plan_layout(h, ps, N, seq).pair_plan &&
        ps.ext_id_slot == h->ext.id_slot) {
        HIP_TRY(hipStreamWaitEvent(st, ps.done, 0));
    } else {
        if (ps.pending) HIP_TRY(hipEventSynchronize(ps.done));       // a prepared plan nobody took: let it finish before the slot is reused
        if (ps.released) HIP_TRY(hipStreamWaitEvent(st, ps.released, 0));      // (the forward before the previous one read this slot)
        if (int rc = enqueue_plan(h, ps, sfm, N, seq, st)) return rc;
    }
    ps.pending = false;
    h->plan_cur ^= 1;
    // From here on kernels that read the slot (and the pinned host words) may be enqueued: whatever way this function is left —
    // a failed launch, a failed workspace reservation, a lane that fails after the other one was launched — the second lane is
    // joined to `st` and ps.released is recorded behind everything, so that a later zett_forward_prepare / zett_forward never
    // rewrites the plan under kernels still in flight.
    struct SlotGuard {
        zett_hypernet* h; zett_hypernet::PlanSlot* ps; hipStream_t st; bool lane_forked = false;
        ~SlotGuard() {
            if (lane_forked && h->lane_stream && h->lane_ev[3]) {
                (void)hipEventRecord(h->lane_ev[3], h->lane_stream);
                (void)hipStreamWaitEvent(st, h->lane_ev[3], 0);
            }
            if (ps->released) (void)hipEventRecord(ps->released, st);
        }
    } slot_guard{h, &ps, st};
    const PlanLayout PL = plan_layout(h, ps, N, seq);
    const PlanArrays& p = PL.p;
    const bool pair_plan = PL.pair_plan;
    if (!h->range_accumulate) HIP_TRY(hipMemsetAsync(h->range_word, 0, 4, st));          // range guard: the word of THIS forward (zett_check_range)
    for (hipEvent_t& e : h->out_ready)
        if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIP_TRY(hipEventSynchronize(ps.done));
    int32_t* hoff = ps.host;
    if (hoff[N + 2] != 0)
        return fail(ZETT_E_INDEX, "surface-form row %d holds an id outside [0, %d) (original_vocab_size %d + %d fallback rows)",
                    hoff[N + 2] - 1, V, c.original_vocab_size, c.n_extra);
    const int64_t Ttot = hoff[N];
    const int D = hoff[N + 1];
    h->stats.packed_tokens = Ttot;
    h->stats.distinct_ids = D;
    h->stats.distinct_positions = Ttot;

    // ---- workspace ------------------------------------------------------------------
    const WorkspaceSizes ws = workspace_sizes(c, sizeof(T), seq, Ttot, D, h->max_chunk_tokens);
    const int64_t MC = ws.chunk_tokens;
    const size_t MCS = (size_t)MC + 768;      // slack rows: the 384-row GEMM tile reads whole tiles of A (per lane, two lanes)
    if (int rc = h->table.reserve(ws.table)) return rc;
    if (int rc = h->x0.reserve(ws.x0)) return rc;
    if (int rc = h->yf.reserve(ws.f32_rows)) return rc;
    if (int rc = h->yt.reserve(ws.lo_rows)) return rc;
    if (int rc = h->big.reserve(ws.big)) return rc;
    if (int rc = h->pre.reserve(ws.f32_rows)) return rc;
    if (int rc = h->ctx.reserve(ws.lo_rows)) return rc;
    if (int rc = h->cf.reserve(ws.f32_rows)) return rc;
    if (int rc = h->ct.reserve(ws.lo_rows)) return rc;
    if (int rc = h->lnstats.reserve(ws.stats)) return rc;
    // LayerNorm fold: in the 16-bit modes the LayerNorms inside the encoder are not launches — the residual GEMM
    // in front writes the 16-bit copy of its fp32 rows and per-row partial statistics, the GEMM behind runs on gamma-folded
    // weights and normalises in its epilogue.  gemm4d only: off when a tile variant is forced.
    const bool fold = h->ln_fold && !std::is_same<T, float>::value && h->gemm_variant == 0 && H % 128 == 0 && H >= 512 &&
                      (int)h->fold_up.size() == c.layers;
    // 16-bit residual stream (r4): with the fold on, the encoder's hidden state travels ONLY as the 16-bit copy of the
    // pre-LayerNorm sum (+ statistics): the producers read their residual rows from it (LN16 epilogue: 4 instead of 10 bytes
    // per element, 16-byte accesses on both sides) and no fp32 sum is written; Zt and Ct alternate as its buffers (Ct is free
    // until the readout).  f16 only by default: the rounding of the stream is 2^-11 per layer there (rel-L2 of the outputs
    // 0.8e-3 -> 1.0e-3 against a tolerance of 2.5e-3), in bf16 2^-8 would leave the tolerance.
    const bool lo_stream = fold && c.layers >= 1 && sizeof(T) == 2 &&
                           (h->residual_lo == 2 || (h->residual_lo == 1 && std::is_same<T, f16_t>::value));
    const size_t ws_parts = fold ? (size_t)(H / 128) * MCS * sizeof(float2) : 0;
    if (int rc = h->lnparts.reserve(ws_parts)) return rc;
    float2* PARTS = h->lnparts.as<float2>();
    float* TBL = h->table.as<float>();
    // 16-bit hoisted table (r6; with the 16-bit residual stream, i.e. f16 by default): the ProjectorBlock of the input projection ends
    // in the LayerNorm-fold PRODUCER the output heads use (dense2 writes the 16-bit copy of its pre-LayerNorm sum + partial row
    // statistics; ln_stats makes (mean, rstd)), and the embeddings' kernel normalises a table row as it reads it: the block's
    // LayerNorm launch is gone (0.40 ms on the headline, 0.16 of XLM-R's 7.3), a table element is 2 bytes on both sides, and a
    // table exchanged between ranks (SURVEY 8e's optional second exchange) would be 239 instead of 478 MB.  One 16-bit rounding of
    // the pre-LayerNorm sum replaces none: the same step the encoder's stream takes per layer.  The buffer keeps its fp32 size:
    // [D, H] 16-bit values, then [D] (mean, rstd).
    const bool table_lo = lo_stream && h->table_lo != 0;
    // (ABI 8) zett_forward_table: the table is the caller's — rows of the GLOBAL distinct-id list in the folded 16-bit layout,
    // computed by zett_table_rows here and on the peer ranks; tok_slot already holds global slots (enqueue_plan)
    const bool ext_table = h->ext.table != nullptr;
    if (ext_table && !table_lo)
        return fail(ZETT_E_INVALID, "zett_forward_table needs the folded 16-bit table: f16 arithmetic with the LayerNorm fold, the 16-bit residual stream and table_lo on");
    T* TBL16 = ext_table ? (T*)const_cast<void*>(h->ext.table) : (T*)h->table.as<float>();
    float* TBLST = ext_table ? const_cast<float*>(h->ext.stats) : (float*)((char*)h->table.as<float>() + (((size_t)D * H * sizeof(T) + 15) / 16) * 16);
    T* X0 = h->x0.as<T>();
    float* Zf = h->yf.as<float>();
    T* Zt = h->yt.as<T>();
    T* BIG = h->big.as<T>();
    float* PRE = h->pre.as<float>();
    T* CTX = h->ctx.as<T>();
    float* Cf = h->cf.as<float>();
    T* Ct = h->ct.as<T>();
    // LayerNorm statistics.  The encoder keeps its hidden state as (pre-LayerNorm sum, statistics, gamma, beta):
    // the LayerNorm kernel writes the 16-bit GEMM operand and the two statistics, and whoever needs the fp32
    // LayerNorm output (the next residual epilogue) recomputes it with ln_affine from the sum it reads anyway.
    // Zf and PRE alternate as the sum buffers, STa and STb as their statistics.
    float* STa = h->lnstats.as<float>();
    float* STb = STa + 2 * MCS;

    Runner<T> R{h, st};
    R.a_rows_readable = (long)MCS;

    // ---- table: input_projection once per distinct source id (A2-A4) -----------------
    if (!ext_table)
        table_phase<T>(R, p.id_list, 0, D, src, src_dtype, MC, MCS, X0, Zf, Zt, BIG, PRE, PARTS, table_lo, TBL16, TBLST, TBL);
    if (R.rc) return R.rc;

    // ---- encoder + heads over row chunks -----------------------------------------------
    // Buffer rows of a chunk are ordered POSITION 0 FIRST (rowops.hip.h chunk_row): rows [0, rows) are position 0 of the
    // chunk's vocabulary rows, the other packed positions follow.  What the last layer and the heads consume — position 0
    // only (modeling_hypernet.py:234) — is then the first `rows` rows of every buffer, with no gather in between.
    const float* lang_vec = lam ? R.Wf("lang_embeddings.weight") + (size_t)lang_index * H : nullptr;
    const float scaling = 1.0f / std::sqrt((float)(H / c.heads));
    // One chunk = vocabulary rows [r0, r1) on one LANE: a stream and a slice of every workspace buffer starting `off` rows in.
    // Normally there is one lane (the caller's stream, offset 0) and the chunks follow each other.  A call that is ONE chunk
    // can instead run as two half-vocabulary chunks on two lanes at once (r4, "concurrent_lanes"): rows are independent, so the
    // two chains of launches interleave on the chip workgroup by workgroup — the last, partly filled round of 256 CUs of one
    // chain's GEMM is filled by the other's, and one chain's prologues / epilogues run under the other's K loops.  That is what
    // a 4 096-row shard of 8 GPUs needs: its N = 4096 launches are 608 tiles = 2.375 rounds.  (Same bits: a row's arithmetic
    // does not depend on the chunk it is in.)  ev_mode: 0 = no completion events, 1 = record out_ready on this lane's stream,
    // 2 = lane 1 of a pair (records lane_ev[1..3]), 3 = lane 0 of a pair (out_ready follows lane 1's events).
    struct Lane { hipStream_t st; size_t off; };
    const size_t ld_parts = MCS;
    auto run_chunk = [&](int64_t r0, int64_t r1, const Lane& lane, int ev_mode) -> int {
        const int rows = (int)(r1 - r0);
        const int tok0 = hoff[r0];
        const int m = hoff[r1] - tok0;
        h->stats.chunks += 1;
        hipStream_t st = lane.st;
        R.st = lane.st;
        R.a_rows_readable = (long)(MCS - lane.off);
        float* const Zf = h->yf.as<float>() + lane.off * H;
        T* const Zt = h->yt.as<T>() + lane.off * H;
        T* const BIG = h->big.as<T>() + lane.off * (size_t)std::max(I, 3 * H);
        float* const PRE = h->pre.as<float>() + lane.off * H;
        T* const CTX = h->ctx.as<T>() + lane.off * H;
        float* const Cf = h->cf.as<float>() + lane.off * H;
        T* const Ct = h->ct.as<T>() + lane.off * H;
        float* const STa = h->lnstats.as<float>() + 2 * lane.off;
        float* const STb = STa + 2 * MCS;
        float2* const PARTS = h->lnparts.as<float2>() + lane.off;

        LnEmbed emb{TBL, p.tok_slot, p.tok_pos, R.Wf("model.embeddings.token_type_embeddings.weight"),
                    R.Wf("model.embeddings.position_embeddings.weight"), lang_vec, seq, p.tok_row, p.row_offset, r0, rows,
                    table_lo ? (const void*)TBL16 : nullptr, TBLST, R.Wf("input_projection.1.ln.weight"), R.Wf("input_projection.1.ln.bias")};
        // hidden state = (sum buffer, statistics, gamma, beta); the embeddings' LayerNorm starts it in (Zf, STb)
        float* hs_sum = Zf;
        float* hs_stats = STb;
        const float* hs_gamma = R.Wf("model.embeddings.LayerNorm.weight");
        const float* hs_beta = R.Wf("model.embeddings.LayerNorm.bias");
        auto embed_ln = [&](const LnEmbed& e, int n, int t0, T* lo, float* stats, float* sum) {
            if constexpr (sizeof(T) == 2) {
                const int tpr8 = h->ln_rows8 ? ln_rows8_tpr(H) : 0;
                if (tpr8) {
                    const dim3 grid8((n + 256 / tpr8 - 1) / (256 / tpr8));
                    if (tpr8 == 32)
                        hipLaunchKernelGGL((layernorm_rows8_kernel<T, true, 32, false>), grid8, dim3(256), 0, st, (const float*)nullptr, H, n, H,
                                           hs_gamma, hs_beta, c.ln_eps_encoder, (float*)nullptr, lo, stats, sum, e, t0, LnReadout{});
                    else
                        hipLaunchKernelGGL((layernorm_rows8_kernel<T, true, 64, false>), grid8, dim3(256), 0, st, (const float*)nullptr, H, n, H,
                                           hs_gamma, hs_beta, c.ln_eps_encoder, (float*)nullptr, lo, stats, sum, e, t0, LnReadout{});
                    R.check("embed_layernorm");
                    return;
                }
            }
            if (H <= 2048)
                hipLaunchKernelGGL((layernorm_rows_kernel<T, true, 64>), dim3((n + 3) / 4), dim3(256), 0, st, (const float*)nullptr, H, n, H,
                                   hs_gamma, hs_beta, c.ln_eps_encoder, (float*)nullptr, lo, stats, sum, e, t0, LnReadout{});
            else
                hipLaunchKernelGGL((layernorm_rows_kernel<T, true, 256>), dim3(n), dim3(256), 0, st, (const float*)nullptr, H, n, H,
                                   hs_gamma, hs_beta, c.ln_eps_encoder, (float*)nullptr, lo, stats, sum, e, t0, LnReadout{});
            R.check("embed_layernorm");
        };
        // Lever 4: the embeddings' output depends on (source id, position) only, so the embeddings' LayerNorm and layer 0's
        // Q/K/V are computed once per DISTINCT pair (P rows instead of m).  The attention kernel reads a packed position's
        // q / k / v through tok_pair, and layer 0's attention-output epilogue adds the residual row of the position's pair
        // (GemmEpilogue::res_index = pair slot per buffer row): until that GEMM has run, the hidden state (operand, sum,
        // statistics) exists per pair only.  Same values, same bits.  Taken when the call is one chunk and at least 15 % of
        // the positions repeat a pair.
        const int P = pair_plan ? hoff[N + 3] : 0;
        const bool pairs = pair_plan && rows == N && P > 0 && (int64_t)P * 100 <= (int64_t)m * 85;
        int32_t* brow_pair = p.tok_pkey;       // (the keys are dead once plan_pairs_kernel has run)
        if (pairs) {
            LnEmbed pe = emb;
            pe.tok_slot = p.pair_tslot; pe.tok_pos = p.pair_pos; pe.tok_row = nullptr;
            embed_ln(pe, P, 0, Zt, hs_stats, lo_stream ? (float*)nullptr : hs_sum);
            hipLaunchKernelGGL(pair_rows_kernel, dim3((m + 255) / 256), dim3(256), 0, st, m, tok0, r0, rows, p, brow_pair);
            R.check("pair_rows");
            h->stats.distinct_positions = P;
        } else {
            embed_ln(emb, m, tok0, Zt, hs_stats, lo_stream ? (float*)nullptr : hs_sum);
        }

        int zrows = m;            // rows of the current hidden state: m, or `rows` (position 0 only) in a position-0-only last layer
        bool raw = false;         // Zt = 16-bit copy of the un-normalised sum (LayerNorm fold) instead of the LayerNorm output
        auto other = [&](float* b) { return b == Zf ? PRE : Zf; };
        auto other_stats = [&](float* b) { return b == STa ? STb : STa; };
        for (int l = 0; l < c.layers && !R.rc; ++l) {
            const std::string lp = "model.encoder.layer." + std::to_string(l) + ".";
            const bool last = l == c.layers - 1;
            const bool cls_only = h->cls_only_last && last;
            // raw: Zt holds the 16-bit copy of the pre-LayerNorm sum (LayerNorm fold: the previous layer's FFN-down wrote it
            // with the statistics in hs_stats) instead of the normalised operand — then this layer's QKV runs on the folded weight
            const T* wqkv = raw ? (const T*)h->fold_qkv[l].w : (const T*)h->qkv_w[l];
            const float* bqkv = raw ? h->fold_qkv[l].b : h->qkv_b[l];
            const float* cqkv = raw ? h->fold_qkv[l].c : nullptr;
            const int64_t waves = attention_waves(rows, H, h->attention_pack != 0).total;
            const int att_flags = (h->attention_fast ? 2 : 0) | (h->attention_pack ? 4 : 0);
            if (!cls_only) {
                const bool by_pair = pairs && l == 0;       // Zt holds the P pair rows; BIG gets their Q/K/V
                GemmEpilogue<T> eq = R.epi();
                eq.bias = bqkv; eq.out_lo = BIG; eq.ld_lo = 3 * H;
                if (raw) { eq.fold_stats = hs_stats; eq.fold_c = cqkv; }
                R.gemm(Zt, H, wqkv, H, by_pair ? P : m, 3 * H, H, eq);
                hipLaunchKernelGGL((attention_rows_kernel<T>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st,
                                   (const T*)BIG, (size_t)3 * H, (const T*)BIG + H, (const T*)BIG + 2 * H, (size_t)3 * H,
                                   H, H / c.heads, p.row_offset, p.row_uniform, p.tok_key, r0, rows, tok0, scaling, 0 | att_flags,
                                   by_pair ? (const int32_t*)p.tok_pair : (const int32_t*)nullptr, CTX);
                R.check("attention");
            } else {
                // Only hidden[:,0] is consumed after this layer (modeling_hypernet.py:234): keys and values for every
                // position, the query (and everything downstream) for position 0 = the first `rows` rows of the
                // hidden state (sum, statistics and 16-bit operand alike).
                T* KV = BIG;                              // [m, 2H]
                T* Q = BIG + (size_t)m * 2 * H;           // [rows, H]  (rows <= m, BIG holds >= m x 3H)
                GemmEpilogue<T> ekv = R.epi();
                ekv.bias = bqkv + H; ekv.out_lo = KV; ekv.ld_lo = 2 * H;
                if (raw) { ekv.fold_stats = hs_stats; ekv.fold_c = cqkv + H; }
                R.gemm(Zt, H, wqkv + (size_t)H * H, H, m, 2 * H, H, ekv);
                GemmEpilogue<T> eq = R.epi();
                eq.bias = bqkv; eq.out_lo = Q; eq.ld_lo = H;
                if (raw) { eq.fold_stats = hs_stats; eq.fold_c = cqkv; }
                R.gemm(Zt, H, wqkv, H, rows, H, H, eq);
                hipLaunchKernelGGL((attention_rows_kernel<T>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st,
                                   (const T*)Q, (size_t)H, (const T*)KV, (const T*)KV + H, (size_t)2 * H,
                                   H, H / c.heads, p.row_offset, p.row_uniform, p.tok_key, r0, rows, tok0, scaling, 1 | att_flags,
                                   (const int32_t*)nullptr, CTX);
                R.check("attention(position 0)");
                zrows = rows;
            }
            // attention output: sum = dense(ctx) + LN(hidden)   (the residual is the LayerNorm of hs_sum, recomputed)
            float* s1 = other(hs_sum);
            float* st1 = other_stats(hs_stats);
            GemmEpilogue<T> eo = R.epi();
            eo.bias = R.Wf(lp + "attention.output.dense.bias");
            if (pairs && l == 0) eo.res_index = brow_pair;       // hs_sum / hs_stats are still per pair here
            if (lo_stream) {
                // the hidden state is Zt: the embeddings' LayerNorm output itself (layer 0: used as stored), or the raw 16-bit
                // sum of the previous layer with its statistics; the new sum goes to Ct
                eo.residual_lo = Zt; eo.ld_res_lo = H;
                if (raw) { eo.res_stats = hs_stats; eo.res_gamma = hs_gamma; eo.res_beta = hs_beta; }
                eo.out_lo = Ct; eo.ld_lo = H; eo.stats_part = PARTS; eo.ld_part = (int)ld_parts;
            } else {
                eo.residual = hs_sum; eo.ld_res = H;
                eo.res_stats = hs_stats; eo.res_gamma = hs_gamma; eo.res_beta = hs_beta;
                eo.out_f32 = s1; eo.ld_f32 = H;
                if (fold) { eo.out_lo = Zt; eo.ld_lo = H; eo.stats_part = PARTS; eo.ld_part = (int)ld_parts; }
            }
            R.gemm(CTX, H, R.Wlo(lp + "attention.output.dense.weight"), H, zrows, H, H, eo);
            const float* g1 = R.Wf(lp + "attention.output.LayerNorm.weight");
            const float* b1 = R.Wf(lp + "attention.output.LayerNorm.bias");
            const T* mid = lo_stream ? Ct : Zt;                  // 16-bit operand of intermediate.dense
            GemmEpilogue<T> ei = R.epi();
            ei.act = ACT_GELU_ERF; ei.out_lo = BIG; ei.ld_lo = I;
            if (fold) {          // the attention-output LayerNorm is folded into intermediate.dense
                R.ln_stats(PARTS, (int)ld_parts, zrows, c.ln_eps_encoder, st1);
                ei.bias = h->fold_up[l].b; ei.fold_stats = st1; ei.fold_c = h->fold_up[l].c;
                R.gemm(mid, H, (const T*)h->fold_up[l].w, H, zrows, I, H, ei);
            } else {
                R.layernorm(s1, zrows, g1, b1, c.ln_eps_encoder, nullptr, Zt, st1);
                ei.bias = R.Wf(lp + "intermediate.dense.bias");
                R.gemm(Zt, H, R.Wlo(lp + "intermediate.dense.weight"), H, zrows, I, H, ei);
            }
            // FFN output: sum = dense(gelu) + LN(s1)
            float* s2 = other(s1);
            float* st2 = other_stats(st1);
            GemmEpilogue<T> ef = R.epi();
            ef.bias = R.Wf(lp + "output.dense.bias");
            if (lo_stream) {      // (also in the last layer: the readout takes the 16-bit sum)
                ef.residual_lo = Ct; ef.ld_res_lo = H; ef.res_stats = st1; ef.res_gamma = g1; ef.res_beta = b1;
                ef.out_lo = Zt; ef.ld_lo = H; ef.stats_part = PARTS; ef.ld_part = (int)ld_parts;
            } else {
                ef.residual = s1; ef.ld_res = H;
                ef.res_stats = st1; ef.res_gamma = g1; ef.res_beta = b1;
                ef.out_f32 = s2; ef.ld_f32 = H;
                if (fold && !last) { ef.out_lo = Zt; ef.ld_lo = H; ef.stats_part = PARTS; ef.ld_part = (int)ld_parts; }
            }
            R.gemm(BIG, I, R.Wlo(lp + "output.dense.weight"), I, zrows, H, I, ef);
            hs_gamma = R.Wf(lp + "output.LayerNorm.weight");
            hs_beta = R.Wf(lp + "output.LayerNorm.bias");
            hs_sum = s2; hs_stats = st2;
            // (the last layer's output LayerNorm is the readout below: position 0 only, whatever the layer computed)
            if (!last) {
                if (fold) { R.ln_stats(PARTS, (int)ld_parts, zrows, c.ln_eps_encoder, st2); raw = true; }
                else R.layernorm(s2, zrows, hs_gamma, hs_beta, c.ln_eps_encoder, nullptr, Zt, st2);
            }
        }
        if (R.rc) return R.rc;

        // position-0 readout + bias head (modeling_hypernet.py:231-234, 260-265) = the last LayerNorm, on the first `rows`
        // buffer rows: Cf = fp32 hidden[:,0] (residual of the heads' ProjectorBlocks), Ct its operand copy, bias head fused.
        // (no encoder layer: the embeddings' LayerNorm is simply taken again for those rows)
        R.layernorm(hs_sum, rows, hs_gamma, hs_beta, c.ln_eps_encoder, Cf, Ct, nullptr,
                    LnReadout{c.predict_bias ? R.Wf("bias_projection.weight") : (const float*)nullptr,
                              c.predict_bias ? R.Wf("bias_projection.bias") : (const float*)nullptr, out_bias + r0},
                    lo_stream ? (const T*)Zt : (const T*)nullptr);
        R.check("readout");
        if (!R.rc) {          // out_bias complete (zett_stream_wait_output)
            if (ev_mode == 2) HIP_TRY(hipEventRecord(h->lane_ev[1], st));
            if (ev_mode == 3) HIP_TRY(hipStreamWaitEvent(st, h->lane_ev[1], 0));
            if (ev_mode == 1 || ev_mode == 3) HIP_TRY(hipEventRecord(h->out_ready[ZETT_OUT_BIAS], st));
        }

        // output heads (modeling_hypernet.py:236-258)
        // LayerNorm fold of the heads (r3): the ProjectorBlock's LayerNorm in front of each final Linear is not a launch either
        const bool fold_heads = fold && h->ln_fold == 1 && h->fold_head_in.w != nullptr;
        {
            if (fold_heads) R.projector("output_projection.0.", Ct, Cf, rows, BIG, PRE, nullptr, CTX, PARTS, (int)ld_parts, STa);
            else R.projector("output_projection.0.", Ct, Cf, rows, BIG, PRE, nullptr, CTX);
            GemmEpilogue<T> e = R.epi();
            e.bias = fold_heads ? h->fold_head_in.b : R.Wf("output_projection.1.bias");
            e.scale = c.rescale ? h->head_scale : (fold_heads ? h->head_one : nullptr);
            e.shift = c.rescale ? h->head_shift : (fold_heads ? h->head_zero : nullptr);
            if (fold_heads) { e.fold_stats = STa; e.fold_c = h->fold_head_in.c; }
            e.out_f32 = out_in + (size_t)r0 * E; e.ld_f32 = E; e.range_final = 1;
            const int width = c.single_head ? EIN : E;
            if (c.single_head && c.separate_out) { e.split_col = E; e.out_f32_b = out_out + (size_t)r0 * E; }
            R.gemm(CTX, H, fold_heads ? (const T*)h->fold_head_in.w : R.Wlo("output_projection.1.weight"), H, rows, width, H, e);
            if (!R.rc) {      // out_in complete: the second head runs behind it
                if (ev_mode == 2) HIP_TRY(hipEventRecord(h->lane_ev[2], st));
                if (ev_mode == 3) HIP_TRY(hipStreamWaitEvent(st, h->lane_ev[2], 0));
                if (ev_mode == 1 || ev_mode == 3) HIP_TRY(hipEventRecord(h->out_ready[ZETT_OUT_IN], st));
            }
        }
        if (c.separate_out && !c.single_head) {
            if (fold_heads) R.projector("output_projection_out.0.", Ct, Cf, rows, BIG, PRE, nullptr, CTX, PARTS, (int)ld_parts, STa);
            else R.projector("output_projection_out.0.", Ct, Cf, rows, BIG, PRE, nullptr, CTX);
            GemmEpilogue<T> e = R.epi();
            e.bias = fold_heads ? h->fold_head_out.b : R.Wf("output_projection_out.1.bias");
            e.scale = c.rescale ? R.Wf("out_scaler.w") : (fold_heads ? h->head_one : nullptr);
            e.shift = c.rescale ? R.Wf("out_scaler.b") : (fold_heads ? h->head_zero : nullptr);
            if (fold_heads) { e.fold_stats = STa; e.fold_c = h->fold_head_out.c; }
            e.out_f32 = out_out + (size_t)r0 * E; e.ld_f32 = E; e.range_final = 1;
            R.gemm(CTX, H, fold_heads ? (const T*)h->fold_head_out.w : R.Wlo("output_projection_out.1.weight"), H, rows, E, H, e);
        }
        return R.rc;
    };
    // two lanes?  Only a call that is one chunk and would not take the pair lever (which needs the whole call in one chunk and is
    // worth more).  auto = the launches of width H (attention output, FFN down: the fewest tiles) would leave more than 8 % of the
    // CU-rounds they occupy idle, on a hypernet wide enough for that to be the cost (H >= 1024), and per-launch timing is off
    // (concurrent launches share the chip: their HIP-event durations overlap and would be counted twice).
    // Measured (r4, same box, Mistral shape): 4 096 rows 9.25 -> 9.11 ms; 8 192 rows 16.14 -> 16.12; 16 384 rows and every full
    // vocabulary slower (28.9 -> 29.8; headline 53.1 -> 55.2 with the pair lever lost): the dispatcher interleaves the two
    // chains' workgroups, but each 256x256 tile still owns its CU, so only the partial last rounds gain.
    bool two_lanes = false;
    if (h->concurrent_lanes && Ttot <= MC && N >= 512) {
        const int Pn = pair_plan ? hoff[N + 3] : 0;
        const bool pairs_taken = pair_plan && Pn > 0 && (int64_t)Pn * 100 <= Ttot * 85;
        const double r = (double)((Ttot + 255) / 256) * (double)((H + 255) / 256) / 256.0;
        two_lanes = h->concurrent_lanes == 2 ||
                    (!pairs_taken && !h->time_gemm && H >= 1024 && r < 4.0 && (std::ceil(r) - r) / std::ceil(r) > 0.08);
    }
    if (two_lanes) {
        if (!h->lane_stream) HIP_TRY(hipStreamCreateWithFlags(&h->lane_stream, hipStreamNonBlocking));
        for (hipEvent_t& e : h->lane_ev)
            if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        // the split is a multiple of 256 rows (the position-0-only last layer and the heads keep their number of row tiles)
        int64_t ra = ((N / 2 + 128) / 256) * 256;
        ra = std::min<int64_t>(std::max<int64_t>(ra, 256), N - 1);
        const size_t off1 = (size_t)(hoff[ra] - hoff[0]) + 384;
        HIP_TRY(hipEventRecord(h->lane_ev[0], st));                       // fork: the plan and the table are complete
        HIP_TRY(hipStreamWaitEvent(h->lane_stream, h->lane_ev[0], 0));
        slot_guard.lane_forked = true;                                    // (joined by the guard on every exit path)
        if (int rc = run_chunk(ra, N, Lane{h->lane_stream, off1}, 2)) return rc;
        if (int rc = run_chunk(0, ra, Lane{st, 0}, 3)) return rc;
    } else {
        int64_t r0 = 0;
        while (r0 < N) {
            int64_t r1 = r0 + 1;
            while (r1 < N && (int64_t)hoff[r1 + 1] - hoff[r0] <= MC) ++r1;
            if (int rc = run_chunk(r0, r1, Lane{st, 0}, r1 == N ? 1 : 0)) return rc;
            r0 = r1;
        }
    }
    R.st = st;
    if (R.rc) return R.rc;
    if (slot_guard.lane_forked) {                      // join now (the guard then has nothing left to join)
        HIP_TRY(hipEventRecord(h->lane_ev[3], h->lane_stream));
        HIP_TRY(hipStreamWaitEvent(st, h->lane_ev[3], 0));
        slot_guard.lane_forked = false;
    }
    h->out_recorded = true;          // (ps.released — the plan slot may be rewritten behind it, zett_forward_prepare — is recorded by the guard)

    if (h->time_gemm) {
        HIP_TRY(hipStreamSynchronize(st));
        double ms = 0.0, fl = 0.0;
        for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, h->ev[i], h->ev[i + 1]) == hipSuccess) { ms += t; fl += h->ev_flops[i / 2]; }
        }
        h->stats.gemm_ms = ms;
        h->stats.gemm_flops_timed = fl;
        for (size_t i = 0; i + 1 < h->ev_used && i / 2 < h->gemm_log.size(); i += 2) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, h->ev[i], h->ev[i + 1]) == hipSuccess) h->gemm_log[i / 2].ms = t;
        }
        if (getenv("ZETT_GEMM_LOG")) {
            for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
                float t = 0.f;
                (void)hipEventElapsedTime(&t, h->ev[i], h->ev[i + 1]);
                const auto& sh = h->ev_shape[i / 2];
                fprintf(stderr, "[zett gemm] M=%6d N=%6d K=%5d tile=%s %8.3f ms %7.1f TF\n", sh[0], sh[1], sh[2],
                        sh[3] == 7 ? "4d " : sh[3] == 8 ? "4dg" : sh[3] == 3 ? "384" : sh[3] == 2 ? "8r " : "128", t, h->ev_flops[i / 2] / (t * 1e9));
            }
        }
    }
    return 0;
}

}  // namespace
