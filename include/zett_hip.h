/*
 * zett_hip.h — C ABI of libzett_hip.so, the MI355X (gfx950) implementation of
 * ZeTT's embedding-prediction hot path.
 *
 * The reference (bminixhofer/zett) has no C ABI on this path: its boundary is a
 * Python class and a Python function.  Each entry point below names the
 * reference interface it stands behind; INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add to hf_hypernet/modeling_hypernet.py and
 * zett/utils.py.
 *
 * Conventions
 *   - plain C types only; no torch / HIP types in any signature.  `stream` is a
 *     hipStream_t passed as void* (NULL = default stream).
 *   - every function returns 0 on success or a negative ZETT_E_* code;
 *     zett_last_error() returns the message of the calling thread's last failure.
 *   - unless a parameter says "host", pointers are device pointers on the
 *     handle's device, owned by the caller (e.g. PyTorch-ROCm tensors).
 *   - a handle is bound to one device; calls on one handle must not overlap.
 */
#ifndef ZETT_HIP_H
#define ZETT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZETT_ABI_VERSION 8      /* 2: zett_stats gained distinct_positions; 3: ZETT_E_RANGE, zett_check_range;
                                   4: ZETT_RETOK_WORDPIECE (zett_retok_model gained piece_continuing, max_input_chars_per_word);
                                   5: zett_forward_prepare, zett_retokenize_async / zett_retok_result;
                                   6: zett_retokenize_async takes NUL-separated text (offsets == NULL); options gemm_tail_split;
                                   7: zett_retok_set_option;
                                   8: zett_table_plan, zett_table_rows, zett_forward_table (the hoisted table shared between ranks) */

enum zett_status {
    ZETT_OK = 0,
    ZETT_E_INVALID = -1,      /* bad argument / unsupported shape                 */
    ZETT_E_HIP = -2,          /* a HIP runtime call failed                        */
    ZETT_E_STATE = -3,        /* weights missing, handle not finalized, ...       */
    ZETT_E_INDEX = -4,        /* surface-form id outside [0, V0 + n_extra)        */
    ZETT_E_NOT_IMPLEMENTED = -5,
    ZETT_E_KEY = -6,          /* character outside the byte table (KeyError)      */
    ZETT_E_RANGE = -7         /* a value left the range of the 16-bit operand type, or an output is not finite
                                 (zett_finalize: a weight; zett_check_range: the last forward)              */
};

/* Bits of the range word (zett_check_range). */
enum zett_range_bits {
    ZETT_RANGE_SOURCE = 1,      /* in_scaler(source_embeddings[id]) of a referenced id does not fit the operand type */
    ZETT_RANGE_ACTIVATION = 2,  /* a 16-bit GEMM output (Q/K/V, FFN intermediate, the operand copy of the residual sum) */
    ZETT_RANGE_OUTPUT = 4,      /* a predicted embedding is inf / NaN                                                */
    ZETT_RANGE_WEIGHT = 8       /* a GEMM weight (or a LayerNorm-folded weight) does not fit: reported by zett_finalize */
};

enum zett_dtype { ZETT_F32 = 0, ZETT_F16 = 1, ZETT_BF16 = 2 };

/* Arithmetic of the dense contractions.
 *   BF16: bf16 operands, fp32 accumulate (v_mfma_f32_32x32x16_bf16); LayerNorm,
 *         softmax, GELU, residual stream and outputs in fp32.  (The reference CLI
 *         defaults to bf16 end to end: scripts/transfer.py:41,145-151.)
 *   F32 : exact fp32 (v_mfma_f32_32x32x2_f32), the hf_hypernet arithmetic.
 *   F16 : IEEE-half operands, fp32 accumulate (v_mfma_f32_*_f16): 8x smaller operand rounding
 *         than BF16 (11 instead of 8 significand bits) at the same MFMA rate; ~4 % slower end to
 *         end, because the GEMMs are power-limited and wider significands cost energy.  Operands
 *         must stay inside the half range (|x| < 65504), which LayerNorm'd activations and
 *         embedding-scale weights do.
 *         The default of the Python layer (zett_amd/hypernet.py DEFAULT_PRECISION): bf16 sits on the edge
 *         of the 1e-2 parity tolerance at the 4096-wide shapes (tests/test_full_size_gpu.py). */
enum zett_precision { ZETT_PREC_BF16 = 0, ZETT_PREC_F32 = 1, ZETT_PREC_F16 = 2 };

/* Shape / flag block.  Mirrors the fields of ZettHypernetConfig that the forward
 * reads (hf_hypernet/configuration_hypernet.py:3-56 plus the fields train.py
 * injects: pad_token_id, original_vocab_size, separate_out_embeddings,
 * hn_n_extra_tokens — SURVEY.md §8a A11). */
typedef struct zett_config {
    int32_t n_embd;               /* E                                             */
    int32_t n_in_embd;            /* E_in = 2E if separate_out_embeddings else E   */
    int32_t hidden;               /* hn_hidden_size                                */
    int32_t intermediate;         /* hn_intermediate_size                          */
    int32_t heads;                /* hn_num_attention_heads                        */
    int32_t layers;               /* hn_n_layers                                   */
    int32_t n_extra;              /* rows of fallback_embeddings = max(X, 1)       */
    int32_t original_vocab_size;  /* V0                                            */
    int32_t pad_token_id;
    int32_t separate_out;         /* separate_out_embeddings                       */
    int32_t single_head;          /* hn_single_head                                */
    int32_t rescale;              /* hn_rescale_embeddings                         */
    int32_t predict_bias;         /* hn_predict_bias                               */
    int32_t embed_lang;           /* hn_embed_lang_id                              */
    int32_t n_langs;
    int32_t max_positions;        /* rows of position_embeddings (514)             */
    float ln_eps_encoder;         /* 1e-5 (roberta-base layer_norm_eps)            */
    float ln_eps_projector;       /* 1e-6 (ProjectorBlock.ln)                      */
} zett_config;

/* Counters of the most recent zett_forward on a handle. */
typedef struct zett_stats {
    int64_t rows;                 /* N                                             */
    int64_t packed_tokens;        /* positions that entered the encoder            */
    int64_t distinct_ids;         /* rows of the hoisted input-projection table    */
    int64_t chunks;               /* encoder row chunks                            */
    double executed_flops;        /* 2*M*N*K summed over every GEMM launch         */
    double gemm_ms;               /* sum of GEMM launch durations (timing on)      */
    int64_t gemm_launches;
    double gemm_flops_timed;      /* flops of the launches counted in gemm_ms      */
    int64_t distinct_positions;   /* rows of layer 0's Q/K/V launch: distinct (source id, position) pairs when that
                                     lever is taken (one chunk, >= 15 % repeats), else packed_tokens            */
} zett_stats;

typedef struct zett_hypernet zett_hypernet;
typedef struct zett_retok zett_retok;

const char* zett_last_error(void);
int zett_abi_version(void);

/* ---- hypernetwork forward -------------------------------------------------
 * Replaces: ZettHypernet.__init__ / load_state_dict / __call__
 *           (hf_hypernet/modeling_hypernet.py:46-154, 156-267). */

int zett_create(const zett_config* cfg, int device, int precision, zett_hypernet** out);
int zett_destroy(zett_hypernet* h);

/* Upload one checkpoint tensor under its PyTorch state_dict name
 * (scripts/convert_to_pt.py:35-49).  `data` may be a host or a
 * device pointer; the library keeps its own repacked copy.  Unknown names and the
 * never-read model.embeddings.word_embeddings.weight are accepted and ignored
 * (returns 0). */
int zett_load_weight(zett_hypernet* h, const char* name, const void* data, int dtype,
                     const int64_t* shape, int ndim);

/* Check that every tensor the config needs was loaded; build fused operands. */
int zett_finalize(zett_hypernet* h);

/* ZettHypernet.__call__(target_surface_forms, source_embeddings=..., lang_index=...)
 *   surface_forms  int32 [n_rows, seq] row-major (what get_surface_form_matrix emits)
 *   source_embeddings [v_src, n_in_embd] row-major, dtype src_dtype
 *   lang_index     -1 = None
 *   out_in  fp32 [n_rows, n_embd]; out_out fp32 [n_rows, n_embd] or NULL when the
 *   config has no second output; out_bias fp32 [n_rows] (zeros when !predict_bias).
 * Returns ZETT_E_INDEX where the reference's F.embedding would raise IndexError. */
int zett_forward(zett_hypernet* h, const int32_t* surface_forms, int64_t n_rows, int32_t seq,
                 const void* source_embeddings, int src_dtype, int64_t v_src, int32_t lang_index,
                 float* out_in, float* out_out, float* out_bias, void* stream);

/* How asynchronous zett_forward is.  The launches of a forward are sized by its PLAN (packed positions, distinct source ids,
 * distinct (id, position) pairs: integers computed on the device from the surface forms), so zett_forward enqueues the plan
 * (six small kernels), waits on the HOST for it and for its error word (ZETT_E_INDEX is returned synchronously, as the
 * reference's IndexError is raised), enqueues the ~60 launches of the forward and returns while they run.  Made on `stream`
 * itself, the plan sits behind whatever that stream still holds — a second zett_forward on the same stream waits for the
 * whole first one before it can enqueue anything.  zett_forward_prepare removes that wait: it enqueues the plan of the NEXT
 * zett_forward (same surface_forms pointer, n_rows, seq) on a stream the handle owns, behind the work `input_stream` holds NOW
 * (the stream the surface forms were produced on — pass one that is not busy with an earlier forward, e.g. the stream the
 * retokenizer ran on before the first forward was enqueued) and returns at once; plans live in two slots that forwards take
 * in turn, so the plan of forward k+1 runs while the kernels of forward k read theirs (it waits for forward k-1 to release
 * its slot).  The matching zett_forward then waits on the host for the plan only.  A prepared plan that the next
 * zett_forward does not match is discarded.  The reference has no counterpart (XLA sizes nothing from data). */
int zett_forward_prepare(zett_hypernet* h, const int32_t* surface_forms, int64_t n_rows, int32_t seq, void* input_stream);

int zett_get_stats(const zett_hypernet* h, zett_stats* out);

/* Per-launch record of the GEMMs of the most recent zett_forward ("time_gemm" on: durations from HIP events recorded on
 * the launch stream around every launch).  What bench.py prices its roofline on, launch class by launch class. */
typedef struct zett_gemm_record {
    int32_t m, n, k;
    int32_t variant;              /* tile kernel that ran: the "gemm_variant" numbering                          */
    int32_t epilogue;             /* bit 0 16-bit output, 1 fp32 output, 2 residual, 3 Rescaler, 4 LayerNorm-fold producer
                                     (16-bit copy + partial statistics), 5 LayerNorm-fold consumer, 6 residual read from the
                                     16-bit stream; bits 8-9 activation
                                     (0 none, 1 tanh-GELU, 2 erf-GELU)                                            */
    float ms;                     /* launch duration (0 when "time_gemm" is off)                                  */
    double flops;                 /* 2*m*n*k                                                                      */
    double bytes;                 /* algorithmic HBM bytes of the launch: A and W read once, every output written
                                     once, the residual rows read once                                            */
} zett_gemm_record;
/* Copies up to `capacity` records into `out` (may be NULL with capacity 0) and stores the number of launches in *count. */
int zett_get_gemm_log(const zett_hypernet* h, zett_gemm_record* out, int64_t capacity, int64_t* count);

/* Range guard of the 16-bit arithmetic modes.  ZETT_PREC_F16 operands overflow to inf above 65504; LayerNorm'd
 * activations and embedding-scale weights stay far inside, but with the LayerNorm fold the operand copy of the RAW
 * residual sum is rounded to half, and a checkpoint with massive activations can leave the range.  Nothing is silent:
 *   - zett_finalize returns ZETT_E_RANGE when a weight (or gamma-folded weight) does not fit the operand type;
 *   - every kernel that writes a 16-bit operand, and the epilogues that write the predicted embeddings, OR a bit of
 *     `zett_range_bits` into a device word that zett_forward clears when it starts.  An inf anywhere upstream reaches
 *     the outputs of its row (inf / NaN propagate through every residual add), so ZETT_RANGE_OUTPUT is the catch-all
 *     and the other bits say where it started;
 *   - zett_check_range waits for `stream`, reads the word of the most recent zett_forward on the handle and returns
 *     ZETT_E_RANGE if it is non-zero (0 otherwise); *flags (may be NULL) receives the word.  zett_forward itself does not
 *     wait for its kernels (see zett_forward_prepare for what it does wait for).  What to do on a hit is the caller's policy: the Python layer (zett_amd/hypernet.py)
 *     re-runs the call with bf16 operands (fp32's exponent range, same MFMA rate) and warns; a hit in BF16 / F32
 *     mode means the outputs are non-finite in the reference's own arithmetic too (non-finite inputs or weights).
 *     With zett_set_option("range_accumulate", 1) zett_forward no longer clears the word and zett_check_range clears it
 *     after reading: several asynchronous forwards, one question at the end (zett_amd/sharding.py, the CLI under torchrun).
 * The reference has no counterpart: its bf16 / fp32 arithmetic cannot leave the range short of inf in fp32. */
int zett_check_range(zett_hypernet* h, void* stream, int32_t* flags);

/* Completion of individual outputs inside a forward, for callers that start moving an output while the rest of the
 * forward still runs (the vocabulary-sharded path starts the all-gather of pred_in under the second head's GEMMs:
 * zett_amd/sharding.py).  zett_forward records an event on its stream when out_bias is complete (after the position-0
 * readout of the last encoder chunk) and when out_in is complete (after the first head's final GEMM of the last chunk);
 * out_out is complete when the forward is, i.e. in stream order.  zett_stream_wait_output makes `stream` (another
 * hipStream_t) wait for that point of the MOST RECENT zett_forward on the handle; it returns at once on the host.
 * The reference has no counterpart (its outputs appear together when XLA's executable returns). */
enum zett_output { ZETT_OUT_IN = 0, ZETT_OUT_BIAS = 1 };
int zett_stream_wait_output(zett_hypernet* h, int which, void* stream);

/* Upper bound of the device bytes zett_forward reserves (and keeps until zett_destroy)
 * for a [n_rows, seq] batch at the current options: the plan's worst case, no pad
 * position and every position a different source id.  The reference has no counterpart
 * (torch's caching allocator owns its activations); callers use it to size --batch_size
 * against free HBM before the first forward. */
int zett_workspace_bytes(const zett_hypernet* h, int64_t n_rows, int32_t seq, int64_t* out_bytes);

/* Options: "max_chunk_tokens" (packed positions per encoder chunk, >= 1024; default: 12 GiB of per-position
 * workspace, at least 131072 positions: 131072 at H = 4096, ~640 k at H = 768), "time_gemm" (0/1: bracket every
 * GEMM launch with HIP events on the launch stream and report the sum in
 * zett_stats.gemm_ms), "cls_only_last_layer" (0/1, default 1), "residual_lo" (0/1/2, default 1: with the LayerNorm fold on, the
 * encoder's hidden state travels as the 16-bit copy of its pre-LayerNorm sum only — the residual GEMMs read their residual rows
 * from it and write no fp32 sum; 1 = in F16 mode (the stream is rounded to 11 significand bits per layer: inside the f16
 * tolerance), 2 = in BF16 mode too (8 bits: outside the bf16 tolerance, A/B only), 0 = fp32 residual stream),
 * "concurrent_lanes" (0/1/2, default 0, A/B only: a call that is one chunk runs as two half-vocabulary chunks on two streams
 * at once — the caller's and one the handle owns, forked behind the hoisted table and joined before zett_forward's work on
 * `stream` ends — so that the partly filled last round of one chain's GEMMs can be filled by the other's; 1 = when the narrowest
 * launches would leave > 8 % of fewer than four CU-rounds idle, the pair lever is not taken and "time_gemm" is off, 2 = always.
 * Measured: 4 096-row shard of the 4096-wide hypernet (2.375 rounds) 8.85 -> 8.82 ms, every larger call slower: each 256x256
 * tile owns its CU, the dispatcher gains only the partial rounds.  Same bits),
 * "attention_fast" (0/1, default 1, A/B only: rows of at most 8 packed positions
 * take the attention kernel's register-resident path), "pair_dedupe" (0/1, default 1: layer 0's
 * Q/K/V once per distinct (source id, position) pair; same bits either way), "ln_fold" (0/1/2, default 1: in the 16-bit
 * modes the LayerNorms inside the encoder AND the ProjectorBlock LayerNorm in front of each output head's final Linear are
 * folded into the GEMMs around them instead of launched; 2 = the encoder's only (A/B); same values to the rounding of the
 * arithmetic, not the same bits; off whenever a tile variant is forced), "gemm_variant"
 * (0 = choose per launch, 1 = 128x128, 2 = 256x256 register-staged eight-wave,
 * 3 = 384x256 LDS-DMA, 7 = 256x256 four-wave direct-to-LDS (the choice for 16-bit
 * operands and K >= "gemm4d_min_k", default 512), 8 = as 7 with the generic epilogue drain; all produce
 * identical bits; a forced variant falls back to 2 where its preconditions do not
 * hold), "gemm_tile_order" (A/B only: 0 = the order in which gemm4d walks column
 * tiles first, default; 1 = row tiles first; same bits), "gemm_group" (A/B only: column — or, order 1, row — tiles per group of that
 * walk, 0 = the default of 4; same bits), "gemm_tail_split" (0/1/2/3, default 1: a 256x256-tile launch whose
 * tiles fill R whole rounds of the 256 CUs and part of one more is cut into 256x256 tiles on the rows of the whole rounds and
 * 128x256 tiles — twice as many workgroups of half the work — on the rest, when that is cheaper (a rank's 4 096-row shard at
 * 8 GPUs: M = 9 682, N = 4096 is 2.375 rounds); 0 = never, 2 = cut every launch in the middle, 3 = 128x256 tiles only (tests);
 * same bits), "table_lo" (0/1, default 1, round 6: with the 16-bit residual stream the hoisted input-projection table is kept as
 * the 16-bit copy of its pre-LayerNorm sum + (mean, rstd) per distinct id and normalised by the embeddings' kernel as it reads a
 * row; 0 = the fp32 table; one 16-bit rounding apart), "ln_rows8" (0/1, default 1, round 6: LayerNorm launches of the 16-bit
 * modes at hidden sizes of 256 k <= 1024 / 512 k <= 2048 give a lane eight columns and a row 32 / 64 lanes; 0 = the float4
 * kernel; the row sums are added in another order, results differ by rounding), "attention_pack" (0/1, default 1, round 6: a
 * last group of 256 / 128 / 64 hidden columns of the attention kernel — hidden size 768 — takes 2 / 4 / 8 rows per wave instead
 * of leaving lanes idle; same bits). */
int zett_set_option(zett_hypernet* h, const char* key, int64_t value);

/* ---- id-affinity row partition (ABI 6) ----------------------------------------------
 * Replaces: the ORDER in which the reference hands target-vocabulary rows to its devices — a random permutation cut into
 * device shards (scripts/transfer.py:54-67, 90-91; zett/utils.py:26).  Rows are independent, so which rank computes which row
 * is free; it decides how many DISTINCT source ids (hoisted input projection) and distinct (id, position) pairs (layer 0's
 * Q/K/V) a rank's shard holds.  zett_partition_rows assigns the n_rows rows of a device-resident surface-form matrix to
 * `world` <= 8 ranks with the given capacities (caps: host array, sum = n_rows) so that rows sharing ids share a rank, and
 * writes to perm_out (device, int32 [n_rows]) the row indices grouped by rank — rank r's rows at [sum(caps[:r]), + caps[r]),
 * ascending.  Deterministic: every rank runs it on the same matrix and gets the same permutation (no communication).  One
 * workgroup, ~2 us per 1024 rows, enqueued on `stream`; `workspace` = zett_partition_workspace_bytes(n_rows, n_ids) device
 * bytes, n_ids = the exclusive upper bound of the ids that count (original_vocab_size + n_extra; others and pad_id are ignored).
 * csrc/partition.hip.h has the algorithm. */
int zett_partition_workspace_bytes(int64_t n_rows, int32_t n_ids, int64_t* out_bytes);
int zett_partition_rows(const int32_t* surface_forms, int64_t n_rows, int32_t seq, int32_t pad_id, int32_t n_ids, int32_t world,
                        const int32_t* caps, int32_t* perm_out, void* workspace, int64_t workspace_bytes, int32_t device, void* stream);
/* dst[order[i]] = src[i] for i < n_rows, rows of row_bytes bytes (4, or a multiple of 16; device pointers; order: int64, < 0 =
 * skip): puts an exchanged block of predicted rows back into vocabulary order — the counterpart of the reference's
 * `preds[indices] += ...` scatter (scripts/transfer.py:96-111).  An HBM stream on `stream`. */
int zett_scatter_rows(const void* src, void* dst, const int64_t* order, int64_t n_rows, int64_t row_bytes, int32_t device, void* stream);

/* ---- the hoisted table shared between ranks (ABI 8) ---------------------------------------
 * Replaces: nothing the reference has — it is SURVEY.md 8e's optional second exchange of the row-sharded prediction
 * (scripts/transfer.py:90-91, zett/utils.py:26): input_projection(in_scaler(source_embeddings[id])) depends on the source id
 * only, so instead of every rank computing it for the distinct ids of ITS rows (8 340 of the 29 187 ids of the whole vocabulary on
 * each of 8 ranks of the headline workload: 2.3 x the work in all), the ranks agree on the distinct ids of the WHOLE vocabulary,
 * each computes 1/P of the table, and the slices are exchanged (the folded 16-bit table: 2 bytes per element + 8 per row).  A table
 * row's bits do not depend on which rank or tile computed it, so the predicted matrices are those of zett_forward bit for bit.
 * f16 arithmetic with the LayerNorm fold, the 16-bit residual stream and "table_lo" on (the defaults); ZETT_E_INVALID otherwise.
 *
 * zett_table_plan: the distinct source ids a forward over `surface_forms` [n_rows, seq] (device int32: every rank passes the WHOLE
 * vocabulary's matrix) would reference, in ascending id order: id_list_out (device int32, capacity original_vocab_size +
 * n_extra) receives them, id_slot_out (device int32 [original_vocab_size + n_extra + 1]) the table slot of every id (exclusive
 * scan of the reference flags), *n_ids_out their number.  Waits for `stream` (one host round trip); ZETT_E_INDEX on an id outside
 * the vocabulary, as zett_forward.
 * zett_table_rows: table rows [first, first + count) of that list — gather + in_scaler + fallback, input_projection.0, the
 * ProjectorBlock up to its pre-LayerNorm sum — into table_out (rows of `hidden` values of the handle's 16-bit type; the buffer
 * holds the WHOLE table, row i at table_out + i * hidden) and stats_out (float [*, 2]: mean, rstd of row i at stats_out + 2 i).
 * Asynchronous on `stream`.
 * zett_forward_table: zett_forward on that table (complete: this rank's rows and its peers') instead of source embeddings. */
int zett_table_plan(zett_hypernet* h, const int32_t* surface_forms, int64_t n_rows, int32_t seq, int32_t* id_slot_out, int32_t* id_list_out,
                    int64_t* n_ids_out, void* stream);
int zett_table_rows(zett_hypernet* h, const int32_t* id_list, int64_t first, int64_t count, const void* source_embeddings, int src_dtype,
                    int64_t v_src, void* table_out, float* stats_out, void* stream);
int zett_forward_table(zett_hypernet* h, const int32_t* surface_forms, int64_t n_rows, int32_t seq, const void* table, const float* table_stats,
                       const int32_t* id_slot, int32_t lang_index, float* out_in, float* out_out, float* out_bias, void* stream);

/* ---- retokenizer ------------------------------------------------------------
 * Replaces: zett.utils.get_surface_form_matrix (zett/utils.py:651-689) and the
 * tokenizers-library Model.tokenize it calls per token (zett/utils.py:681): BPE, Unigram and WordPiece models. */

enum zett_retok_kind { ZETT_RETOK_BPE = 0, ZETT_RETOK_UNIGRAM = 1, ZETT_RETOK_WORDPIECE = 2 };

/* Host-side description of the hn tokenizer's bare model.  Pieces are given as RAW
 * BYTES (byte-level token strings already mapped through the reference's
 * CHARS_TO_BYTES table, zett/utils.py:351-609); pieces that contain a character
 * outside that table can never match a byte-level token and must be omitted. */
typedef struct zett_retok_model {
    int32_t kind;                  /* zett_retok_kind                              */
    int32_t n_pieces;
    const uint8_t* piece_bytes;    /* host: concatenated piece bytes               */
    const int32_t* piece_offsets;  /* host: n_pieces + 1                           */
    const int32_t* piece_ids;      /* host: vocabulary id of each piece            */
    const double* piece_scores;    /* host: Unigram log-probs (NULL for BPE)       */
    double unigram_min_score;      /* Unigram: min score over the WHOLE vocabulary (also pieces
                                      omitted above); unknown pieces score min - 10 */
    int32_t n_merges;              /* BPE                                          */
    const int32_t* merges;         /* host: n_merges x 3 (left id, right id, new id), rank = row */
    int32_t unk_id;                /* -1 = none                                    */
    int32_t fuse_unk;              /* BPE fuse_unk (Unigram always fuses)          */
    int32_t byte_fallback;         /* BPE/Unigram byte_fallback                    */
    const int32_t* byte_fallback_ids; /* host: 256 ids of "<0xXX>" tokens, -1 = absent (may be NULL) */
    int32_t ignore_merges;         /* BPE ignore_merges                            */
    int32_t n_special;
    const uint8_t* special_bytes;  /* host: hn tokenizer all_special_tokens, raw bytes */
    const int32_t* special_offsets;/* host: n_special + 1                          */
    const int32_t* special_ids;    /* host                                          */
    /* WordPiece (tokenizers WordPiece::tokenize; zett/tokenizer_converters.py:370-373 carries such hn tokenizers through).
     * A lookup at the start of a word matches a vocabulary entry as listed; a lookup further in matches
     * continuing_subword_prefix + substring.  The caller lists every entry once as it is (piece_continuing 0) and every
     * entry that starts with the prefix once more with the prefix stripped (piece_continuing 1) — with the empty prefix
     * that convert_to_byte_level leaves, each entry twice.  unk_id = id of unk_token, -1 when it is not in the vocabulary
     * (a word that needs it then fails the call with ZETT_E_STATE, as the library raises). */
    const uint8_t* piece_continuing;   /* host: n_pieces flags (may be NULL: no continuing pieces); WordPiece only */
    int32_t max_input_chars_per_word;  /* WordPiece: longer words are [UNK] (library default 100)                  */
} zett_retok_model;

int zett_retok_create(const zett_retok_model* model, int device, zett_retok** out);
int zett_retok_destroy(zett_retok* r);
/* A/B switches of a retokenizer handle (as zett_set_option for the forward): "unigram_workgroup" — which stage-2 kernel Unigram
 * models run: the workgroup-per-64-tokens kernel (all piece lookups of 64 tokens in flight at once, then the Viterbi walk on
 * LDS) or the lane-per-token kernel every model kind used before.  1 (default) = by size (the workgroup kernel for calls of up to
 * 32 768 tokens), 2 = always the workgroup kernel, 0 = never.  Same ids either way (tests/test_retok_gpu.py).  Unknown key:
 * ZETT_E_INVALID.  No counterpart in the reference (tokenizers' Unigram::tokenize has one code path). */
int zett_retok_set_option(zett_retok* r, const char* key, int64_t value);

/* get_surface_form_matrix(tokens, maxlen, tokenizer_to_use)
 *   token_chars   device: UTF-8 text of the byte-level target tokens, concatenated
 *   offsets       device: int32 [n_tokens + 1] byte offsets into token_chars
 *   out           device: int32 [n_tokens, maxlen], pre-filled here with pad_id
 *   n_truncated   host out: number of tokens cut to maxlen
 * The byte-table lookup (CHARS_TO_BYTES) runs on the device; a character outside
 * the table returns ZETT_E_KEY with the offending token index in *bad_token
 * (reference: KeyError at zett/utils.py:675). */
int zett_retokenize(zett_retok* r, const uint8_t* token_chars, const int32_t* offsets,
                    int64_t n_tokens, int32_t maxlen, int32_t pad_id, int32_t* out,
                    int64_t* n_truncated, int64_t* bad_token, void* stream);

/* The same without the two host round trips of zett_retokenize (which reads the text length, offsets[n_tokens], back before it
 * launches, and the counts after): the caller passes n_text — it built the text — and the call returns as soon as its six
 * kernels are enqueued on `stream`.  Counts and errors are collected by zett_retok_result, ONCE for all asynchronous calls
 * since the previous query: it waits for the last of them, sums their truncated tokens into *n_truncated, and on a failure
 * returns what zett_retokenize would (ZETT_E_KEY / ZETT_E_STATE) for the EARLIEST failing call, whose ordinal since the
 * previous query goes to *bad_call and whose token index to *bad_token (both nullable; -1 when nothing failed).  The id
 * matrices of the calls before the failing one are complete.  A synchronous zett_retokenize in between discards what the
 * asynchronous calls before it reported; at most 2^20 calls may be outstanding between two queries (ZETT_E_STATE beyond).
 * offsets == NULL (ABI 6): token_chars holds the n_tokens tokens NUL-SEPARATED (n_text bytes, n_tokens - 1 of them NUL — exactly
 * what "\0".join(tokens).encode() gives; no byte-level token holds a NUL, byte 0 being written U+0100): the token boundaries
 * are found on the device by the scan that numbers the characters, so the host does no per-token work at all.  A text with
 * another number of separators is reported as ZETT_E_INVALID by zett_retok_result.
 * LIFETIME: `offsets` of every outstanding call must stay allocated and unchanged until zett_retok_result returns — on a
 * ZETT_E_KEY it is read back to name the failing token (zett_amd.surface_forms.DeviceRetokenizer holds the tensors).  A call
 * that fails while it enqueues is taken back (it does not count as outstanding).  The reference has no counterpart (its loop is host code). */
int zett_retokenize_async(zett_retok* r, const uint8_t* token_chars, const int32_t* offsets, int64_t n_tokens, int64_t n_text,
                          int32_t maxlen, int32_t pad_id, int32_t* out, void* stream);
int zett_retok_result(zett_retok* r, int64_t* n_truncated, int64_t* bad_call, int64_t* bad_token);

/* ---- training use of the forward ---------------------------------------------------------
 * Replaces: the hypernetwork forward inside the loss of the reference's train_step / eval_step (train.py:1007-1013,
 * 1191-1197: state.apply_fn({"params": params["hypernet"]}, target_surface_forms, target_priors, source_embeddings,
 * lang_index) under jax.value_and_grad) — i.e. the same forward, differentiable with respect to every hypernetwork
 * parameter.  First slice: fp32 arithmetic in the reference's as-written dense [N, L', H] layout.  The primitives below
 * are what zett_amd/autograd.py (a torch.autograd.Function: torch holds the tensors and the tape, nothing else) builds
 * the forward that keeps its activations and the backward from.  All pointers are device pointers, fp32, row-major;
 * `stream` is a hipStream_t.  Dense contractions — forward, dgrad, wgrad — are ONE entry point on the library's TN
 * GEMM family (fp32 MFMA): dgrad is the same contraction against the transposed weight, wgrad the same contraction of
 * the two transposed activations (zett_op_transpose_f32 zero-pads the row count to the 32-wide K step). */
/* out[m, n] = act(a[m, :] . w[n, :] + bias[n]) + residual[m, n]   (bias, residual nullable; act: 0 none, 1 tanh-GELU, 2 erf-GELU;
 * k % 32 == 0, lda / ldw % 4 == 0) */
int zett_op_gemm_f32(const float* a, int32_t lda, const float* w, int32_t ldw, int64_t m, int32_t n, int32_t k, const float* bias, int32_t act,
                     const float* residual, int32_t ld_res, float* out, int32_t ld_out, void* stream);
/* The same contraction on 16-bit MFMA operands (prec = ZETT_PREC_BF16 | ZETT_PREC_F16), fp32 accumulate, fp32 epilogue and output
 * (k % 64 == 0, lda / ldw % 8 == 0): the tile kernels of the inference path.  Operands are made by zett_op_convert_lo
 * (out[r, c] = lo(in[r, c]), columns zero-padded to cols_padded) and zett_op_transpose_lo (out[c, r] = lo(in[r, c]), rows
 * zero-padded to rows_padded): the conversion is fused with the layout change dgrad / wgrad need anyway. */
int zett_op_gemm_lo(int32_t prec, const void* a, int32_t lda, const void* w, int32_t ldw, int64_t m, int32_t n, int32_t k, const float* bias, int32_t act,
                    const float* residual, int32_t ld_res, float* out, int32_t ld_out, void* stream);
int zett_op_convert_lo(int32_t prec, const float* in, int32_t ld_in, void* out, int32_t ld_out, int64_t rows, int32_t cols, int32_t cols_padded, void* stream);
int zett_op_transpose_lo(int32_t prec, const float* in, int32_t ld_in, void* out, int32_t ld_out, int64_t rows, int32_t cols, int64_t rows_padded, void* stream);
/* The transposed operand of an activation that is already stored as a 16-bit operand of type prec (no conversion). */
int zett_op_transpose_lo16(int32_t prec, const void* in, int32_t ld_in, void* out, int32_t ld_out, int64_t rows, int32_t cols, int64_t rows_padded, void* stream);
/* Everything a Linear's backward needs from its output gradient dy [rows, cols], in one read: dy_lo = lo(dy) (dgrad's A operand),
 * dy_t [cols, rows_padded] = lo(dy)^T (wgrad's A operand, rows zero-padded), colsum_part [ceil(rows / 64), cols] = the column sums
 * of each 64-row band (their sum over the bands is the bias gradient).  act_z (nullable): the Linear's output went through a GELU
 * (act_kind 1 tanh, 2 erf) and dy is the gradient of the GELU's OUTPUT: dy * gelu'(act_z) is formed on the fly and used for
 * all three results (the activation's backward costs no pass of its own). */
int zett_op_grad_operands_lo(int32_t prec, const float* dy, int32_t ld, const float* act_z, int32_t ld_z, int32_t act_kind, int64_t rows, int32_t cols,
                             int64_t rows_padded, void* dy_lo, int32_t ld_lo, void* dy_t, int32_t ld_t, float* colsum_part, void* stream);
/* out[c, r] = in[r, c] (r < rows), 0 for rows <= r < rows_padded */
int zett_op_transpose_f32(const float* in, int32_t ld_in, float* out, int32_t ld_out, int64_t rows, int32_t cols, int64_t rows_padded, void* stream);
/* out[c] (+)= sum_r in[r, c] */
int zett_op_colsum_f32(const float* in, int32_t ld, int64_t rows, int32_t cols, float* out, int32_t accumulate, void* stream);
/* op 0: a + b; 1: a * b; 2: a * vec[col] + vec2[col] (vec NULL: 1, vec2 NULL: 0); 3: a + s[row] * vec[col]; 4: a * s[row] (a NULL: s[row] * vec[col]) */
int zett_op_elementwise_f32(int32_t op, const float* a, const float* b, const float* vec, const float* vec2, const float* s, float* out,
                            int64_t n, int32_t cols, void* stream);
/* out[r] = a[r, :] . w + b[0] (b nullable) */
int zett_op_rowdot_f32(const float* a, int32_t ld, const float* w, const float* b, float* out, int64_t rows, int32_t cols, void* stream);
/* y = LayerNorm(x) (two-pass variance); stats[r] = (mean, rstd); y_lo (nullable, type prec): the same values as the 16-bit
 * operand of the next contraction.  4 <= h <= 8192, h % 4 == 0. */
int zett_op_layernorm_fwd_f32(const float* x, int32_t ld, const float* gamma, const float* beta, float eps, float* y, float* stats,
                              int64_t rows, int32_t h, void* y_lo, int32_t prec, void* stream);
/* dx for the gradient dy (+ dy2, nullable: the part arriving over the residual branch), and the parameter gradients as
 * n_part partial sums: partials [n_part, 2, h] — partials[:, 0].sum(0) = dgamma, partials[:, 1].sum(0) = dbeta (workgroup b
 * walks rows b, b + n_part, ...: deterministic for a given n_part).  4 <= h <= 8192, h % 4 == 0. */
int zett_op_layernorm_bwd_f32(const float* dy, const float* dy2, const float* x, int32_t ld, const float* stats, const float* gamma, float* dx,
                              float* partials, int32_t n_part, int64_t rows, int32_t h, void* stream);
int zett_op_gelu_fwd_f32(const float* z, float* h, int64_t n, int32_t kind, void* stream);
int zett_op_gelu_bwd_f32(const float* z, const float* dh, float* dz, int64_t n, int32_t kind, void* stream);
/* h_lo = lo(gelu(z)): the activation as the next contraction's 16-bit operand (its fp32 value is never stored) */
int zett_op_gelu_fwd_lo(int32_t prec, const float* z, void* h_lo, int64_t n, int32_t kind, void* stream);
/* softmax(q k^T / sqrt(d) + finfo.min * !mask) v per vocabulary row and head (eager semantics: a row whose keys are all
 * masked attends uniformly).  The positions of row n are rows [row_offset[n], row_offset[n+1]) of k / v (packed: only the
 * positions the row keeps) or [n*seq, (n+1)*seq) when row_offset is NULL (the reference's dense layout); at most seq <= 32
 * positions per row; mask[t] = position t is visible as a key.  cls_only: one query per row (position 0), q and ctx hold
 * one row per vocabulary row (the position-0-only last layer).  probs [n_rows, heads, seq, seq] is kept for the backward.
 * ctx_lo (nullable, type prec, same leading dimension): the context is written as the 16-bit operand of the contraction behind it
 * INSTEAD of fp32 (ctx may then be NULL).
 * Head dims up to 256 (the backward: above 128 for rows of up to 16 positions — the row's keys, values and their gradients
 * live in registers). */
int zett_op_attention_fwd_f32(const float* q, int32_t ldq, const float* k, const float* v, int32_t ld, const uint8_t* mask, const int32_t* row_offset,
                              int64_t n_rows, int32_t seq, int32_t heads, int32_t head_dim, int32_t cls_only, float* ctx, int32_t ld_ctx, float* probs,
                              void* ctx_lo, int32_t prec, void* stream);
int zett_op_attention_bwd_f32(const float* dctx, int32_t ld_ctx, const float* q, int32_t ldq, const float* k, const float* v, int32_t ld, const float* probs,
                              const int32_t* row_offset, int64_t n_rows, int32_t seq, int32_t heads, int32_t head_dim, int32_t cls_only, float* dq, int32_t ld_dq,
                              float* dk, float* dv, int32_t ld_d, void* stream);
/* out[r, :] = a[r, :] + src[idx[r], :] (a NULL: 0);   dst[idx[r], :] += src[r, :] (atomic) */
int zett_op_gather_rows_f32(const float* a, const float* src, int32_t ld_src, const int32_t* idx, float* out, int64_t rows, int32_t cols, void* stream);
int zett_op_scatter_add_rows_f32(float* dst, int32_t ld_dst, const int32_t* idx, const float* src, int64_t rows, int32_t cols, void* stream);
/* A2 + A3 (modeling_hypernet.py:170-188) per position, and its backward: dfallback accumulated in place (zero it first),
 * prod = dx * source row and keep = dx on source rows (0 on fallback rows): their column sums are d in_scaler.w / d in_scaler.b */
int zett_op_gather_fwd_f32(const int32_t* ids, int64_t n_tokens, const void* src, int32_t src_dtype, int32_t e_in, int32_t v0, const float* fallback,
                           const float* sw, const float* sb, float* x, void* stream);
int zett_op_gather_bwd_f32(const int32_t* ids, int64_t n_tokens, const void* src, int32_t src_dtype, int32_t e_in, int32_t v0, const float* dx,
                           float* dfallback, float* prod, float* keep, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ZETT_HIP_H */
