// retok.hip.h — get_surface_form_matrix on the device (reference zett/utils.py:651-689).
//
// Two stages, both HBM/latency-bound integer work (no MFMA here):
//
//  1. chars -> bytes.  The target tokens arrive as UTF-8 text of byte-level characters
//     (the GPT-2 printable-character alphabet, zett/utils.py:351-609).  A workgroup
//     takes 4 KiB of text with coalesced 16-byte loads, marks character starts, looks
//     every character up in the byte table staged in LDS, and compacts the bytes with a
//     wavefront scan (DPP-free shuffles inside a wave, LDS across the 4 waves); block
//     totals are chained by one tiny scan launch.  A character outside the table is the
//     reference's KeyError (zett/utils.py:675).
//
//  2. bytes -> source-subtoken ids, one lane per token: special-token lookup
//     (zett/utils.py:671-673) or the hn tokenizer's bare model (zett/utils.py:681):
//       BPE      tokenizers `BPE::tokenize`: merge_word (byte fallback / unk / drop) then
//                Word::merge_all with its (rank, position) priority order and its
//                expired-entry rule, on open-addressing pair tables;
//       Unigram  tokenizers `Unigram::tokenize`: Viterbi over an FNV-hashed piece table
//                (strictly-greater updates, start positions ascending => earliest start
//                wins ties), unknown runs fused, optional byte fallback;
//       WordPiece tokenizers `WordPiece::tokenize` (zett/tokenizer_converters.py:370-373 carries
//                WordPiece hn tokenizers through): greedy longest match from every start,
//                continuing pieces in their own key space of the same hash table, the whole
//                word [UNK] on a miss or beyond max_input_chars_per_word;
//     then truncation to maxlen with the n_truncated count (zett/utils.py:683-685).
//
// Results are integers: bit-exact against the oracle and the reference by construction.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.hip.h"
#include "rowops.hip.h"

namespace zett {

struct PieceEntry {
    uint64_t hash;
    int32_t off;
    int32_t len;      // 0 = empty slot
    int32_t id;
    int32_t where;    // WordPiece: 0 = matches at the start of a word, 1 = continuing piece (prefix stripped); else 0
    double score;
};

struct MergeEntry {
    int32_t a, b;     // a == -1: empty slot
    int32_t rank, new_id;
};

struct RetokTables {
    const PieceEntry* pieces; uint32_t piece_mask; const uint8_t* piece_blob;
    const uint32_t* piece_bits; uint32_t piece_bits_mask;      // (r6) one bit per piece hash, 32 bits of table per piece: see piece_maybe

    const PieceEntry* specials; uint32_t special_mask; const uint8_t* special_blob;
    const MergeEntry* merges; uint32_t merge_mask;
    const int32_t* single_id;     // [256] id of the one-byte piece or -1
    const int32_t* bf_ids;        // [256] id of "<0xXX>" or -1
    const int16_t* cp_to_byte;    // [324] code point -> byte or -1
    int kind, unk_id, fuse_unk, byte_fallback, ignore_merges, max_piece_len, max_word_chars;
    double unk_score;             // min_score - kUnkPenalty
#ifdef ZETT_RETOK_DEBUG
    int debug_stop;               // tools/retok_phases.sh only (-DZETT_RETOK_DEBUG): the stage-2 kernels return after phase <n>; the library never defines it
#endif
};
#ifdef ZETT_RETOK_DEBUG
#define RETOK_STOP(n) do { if (t.debug_stop == (n)) return; } while (0)
#else
#define RETOK_STOP(n) do { } while (0)
#endif

constexpr uint64_t FNV_OFFSET = 14695981039346656037ull;
constexpr uint64_t FNV_PRIME = 1099511628211ull;
constexpr uint64_t FNV_OFFSET_CONT = FNV_OFFSET ^ 0x5bd1e9955bd1e995ull;      // key space of WordPiece continuing pieces

__host__ __device__ inline uint64_t fnv_step(uint64_t h, uint8_t b) { return (h ^ b) * FNV_PRIME; }
__host__ __device__ inline uint32_t piece_slot(uint64_t h, uint32_t mask) { return (uint32_t)(h ^ (h >> 32)) & mask; }
// Pre-filter of the Unigram lookups: most (start, end) substrings of a token are not pieces, and each of those misses was a round
// trip to a 16-64 MB table.  A bitmap with 32 bits per piece (XLM-R: 1 MB — L2-resident) answers ~97 % of them from L2: bit
// piece_bit(h) is set for every piece's hash; a clear bit is a certain miss, a set bit goes to the table as before.  Same ids.
__host__ __device__ inline uint32_t piece_bit(uint64_t h, uint32_t mask) { return (uint32_t)((h * 0xD6E8FEB86659FD93ull) >> 37) & mask; }
__device__ inline bool piece_maybe(const uint32_t* bits, uint32_t mask, uint64_t h) {
    if (!bits) return true;
    const uint32_t b = piece_bit(h, mask);
    return (bits[b >> 5] >> (b & 31)) & 1u;
}
__host__ __device__ inline uint32_t merge_slot(int32_t a, int32_t b, uint32_t mask) {
    const uint64_t k = (((uint64_t)(uint32_t)a) << 32 | (uint32_t)b) * 0x9E3779B97F4A7C15ull;
    return (uint32_t)(k >> 32) & mask;
}

// bytes [0, len) of a piece in the global blob against the candidate substring.  The blob side is fetched as WHOLE 8-byte words,
// all of them before the first comparison (r6): the byte loop this replaces left after the first mismatch — so every byte was a
// load the previous one's comparison had to wait for, up to len dependent L2 round trips behind a hash hit.  The blob carries 16
// bytes of slack behind its last piece (build_piece_table), its base is 256-byte aligned (hipMalloc).
constexpr int PIECE_CMP_WORDS = 4;          // pieces of up to 32 bytes take the word path
template <typename BP>
__device__ inline bool piece_bytes_equal(const uint8_t* p, BP s, int len) {
    if (len > 8 * PIECE_CMP_WORDS) {
        int i = 0;
        while (i < len && p[i] == s[i]) ++i;
        return i == len;
    }
    const uint64_t* base = (const uint64_t*)((uintptr_t)p & ~(uintptr_t)7);
    const int sh = (int)((uintptr_t)p & 7) * 8;
    const int nw = (len + 7) >> 3;
    uint64_t w[PIECE_CMP_WORDS + 1];
#pragma unroll
    for (int j = 0; j <= PIECE_CMP_WORDS; ++j) w[j] = j <= nw ? base[j] : 0ull;
    bool eq = true;
#pragma unroll
    for (int j = 0; j < PIECE_CMP_WORDS; ++j) {
        if (j < nw) {
            const uint64_t have = sh ? ((w[j] >> sh) | (w[j + 1] << (64 - sh))) : w[j];
            uint64_t want = 0;
            const int nb = len - 8 * j < 8 ? len - 8 * j : 8;
            for (int b = 0; b < nb; ++b) want |= (uint64_t)s[8 * j + b] << (8 * b);
            const uint64_t mask = nb == 8 ? ~0ull : ((1ull << (8 * nb)) - 1ull);
            eq = eq && ((have ^ want) & mask) == 0;
        }
    }
    return eq;
}

// BP: pointer to the token's bytes — global memory, or the wave's staged text in LDS (stage 2 below)
template <typename BP>
__device__ inline const PieceEntry* piece_find(const PieceEntry* tab, uint32_t mask, const uint8_t* blob, uint64_t h, BP s, int len) {
    for (uint32_t slot = piece_slot(h, mask);; slot = (slot + 1) & mask) {
        const PieceEntry* e = tab + slot;
        if (e->len == 0) return nullptr;
        if (e->hash == h && e->len == len) {
            if (piece_bytes_equal(blob + e->off, s, len)) return e;
        }
    }
}

// the same lookup when the entry of the FIRST slot has already been fetched (unigram_token fetches several at once)
template <typename BP>
__device__ inline const PieceEntry* piece_find_from(const PieceEntry* tab, uint32_t mask, const uint8_t* blob, uint64_t h, BP s, int len,
                                                    uint32_t slot, const PieceEntry& first) {
    if (first.len == 0) return nullptr;
    if (first.hash == h && first.len == len) {
        if (piece_bytes_equal(blob + first.off, s, len)) return tab + slot;
    }
    for (slot = (slot + 1) & mask;; slot = (slot + 1) & mask) {
        const PieceEntry* e = tab + slot;
        if (e->len == 0) return nullptr;
        if (e->hash == h && e->len == len) {
            if (piece_bytes_equal(blob + e->off, s, len)) return e;
        }
    }
}

// WordPiece: the same table holds word-initial pieces (where 0, hashed from FNV_OFFSET) and continuing pieces (where 1,
// hashed from FNV_OFFSET_CONT): `where` is part of the key
template <typename BP>
__device__ inline const PieceEntry* piece_find_where(const PieceEntry* tab, uint32_t mask, const uint8_t* blob, uint64_t h, BP s, int len, int where) {
    for (uint32_t slot = piece_slot(h, mask);; slot = (slot + 1) & mask) {
        const PieceEntry* e = tab + slot;
        if (e->len == 0) return nullptr;
        if (e->hash == h && e->len == len && e->where == where) {
            if (piece_bytes_equal(blob + e->off, s, len)) return e;
        }
    }
}

__device__ inline const MergeEntry* merge_find(const RetokTables& t, int32_t a, int32_t b) {
    if (t.merge_mask == 0xffffffffu) return nullptr;   // no merges at all
    for (uint32_t slot = merge_slot(a, b, t.merge_mask);; slot = (slot + 1) & t.merge_mask) {
        const MergeEntry* e = t.merges + slot;
        if (e->a == -1) return nullptr;
        if (e->a == a && e->b == b) return e;
    }
}

// ---------------------------------------------------------------------------------------
// stage 1: UTF-8 byte-level characters -> raw bytes
// ---------------------------------------------------------------------------------------
constexpr int CH_PER_THREAD = 16;
constexpr int CH_PER_BLOCK = 256 * CH_PER_THREAD;

__device__ inline int block_excl_scan_256(int v, int* total, int* lds /* [4] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int n = __shfl_up(inc, off, 64);
        if (lane >= off) inc += n;
    }
    __syncthreads();
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += lds[w];
    *total = lds[0] + lds[1] + lds[2] + lds[3];
    return base + inc - v;
}

// MODE 0: count character starts per block.  MODE 1: decode + compact.
// SEP (r5): the text holds the tokens NUL-SEPARATED instead of coming with an offsets array (zett_retokenize_async with
// offsets == nullptr: what one "\0".join(tokens).encode() on the host produces — no per-token work there at all).  A NUL is
// then not a character: it emits no byte, and the k-th one marks the start of token k + 1 — the same block scan that numbers the
// characters numbers the separators, and raw_off[] (first raw byte of every token, what stage 2 consumes) is written HERE
// instead of by token_raw_offsets_kernel.  blk_count then holds two arrays of n_blocks + 1 ints ... see the launcher.  A bad
// character is reported with its TOKEN index (the separators before it) instead of its text position.
template <int MODE, bool SEP>
__global__ __launch_bounds__(256) void chars_to_bytes_kernel(const uint8_t* __restrict__ text, int64_t n_text,
                                                             const int16_t* __restrict__ cp_to_byte,
                                                             int32_t* __restrict__ blk_count,      // MODE 0 out / MODE 1 in (scanned)
                                                             uint8_t* __restrict__ raw, uint32_t* __restrict__ raw_pos,
                                                             unsigned long long* __restrict__ err_pos, uint32_t call,
                                                             int32_t* __restrict__ sep_count = nullptr,   // SEP: MODE 0 out / MODE 1 in (scanned)
                                                             int32_t* __restrict__ raw_off = nullptr, int64_t n_tokens = 0) {
    __shared__ int16_t s_tab[324];
    __shared__ int s_red[4];
    __shared__ int s_red2[4];
    __shared__ uint8_t s_next[256];        // first byte of the following thread's span
    if (MODE == 1)
        for (int i = threadIdx.x; i < 324; i += 256) s_tab[i] = cp_to_byte[i];
    const int64_t base = (int64_t)blockIdx.x * CH_PER_BLOCK + (int64_t)threadIdx.x * CH_PER_THREAD;
    uint8_t b[CH_PER_THREAD + 1];
    if (base + CH_PER_THREAD <= n_text && ((uintptr_t)(text + base) & 15) == 0) {
        const uint4 v = *(const uint4*)(text + base);
        memcpy(b, &v, 16);
    } else {
#pragma unroll
        for (int i = 0; i < CH_PER_THREAD; ++i) b[i] = (base + i < n_text) ? text[base + i] : 0x80;   // 0x80: never a start
    }
    if (MODE == 1) {
        s_next[threadIdx.x] = b[0];
        __syncthreads();
        if (threadIdx.x < 255) b[CH_PER_THREAD] = s_next[threadIdx.x + 1];
        else b[CH_PER_THREAD] = (base + CH_PER_THREAD < n_text) ? text[base + CH_PER_THREAD] : 0;
    }
    int cnt = 0, nsep = 0;
#pragma unroll
    for (int i = 0; i < CH_PER_THREAD; ++i) {
        const bool in = base + i < n_text;
        cnt += in && ((b[i] & 0xC0) != 0x80) && !(SEP && b[i] == 0);
        if (SEP) nsep += in && b[i] == 0;
    }
    int total = 0, total_sep = 0;
    const int excl = block_excl_scan_256(cnt, &total, s_red);
    int tok = 0;
    if (SEP) tok = block_excl_scan_256(nsep, &total_sep, s_red2);
    if (MODE == 0) {
        if (threadIdx.x == 0) { blk_count[blockIdx.x] = total; if (SEP) sep_count[blockIdx.x] = total_sep; }
        return;
    }
    uint32_t pos = (uint32_t)blk_count[blockIdx.x] + (uint32_t)excl;
    bool sep_ok = true;
    if (SEP) {
        tok += sep_count[blockIdx.x];                              // separators before this thread's span = index of the token it starts in
        // n_tokens tokens are n_tokens - 1 separators: a text that holds another number is refused (raw_off stays all zero — the
        // launcher cleared it — so stage 2 sees empty tokens, and the result query reports ZETT_E_INVALID)
        sep_ok = sep_count[gridDim.x] == (int32_t)(n_tokens - 1);
        if (!sep_ok && base == 0) atomicMin(err_pos, ((unsigned long long)call << 32) | 0xfffffffeull);
    }
#pragma unroll
    for (int i = 0; i < CH_PER_THREAD; ++i) {
        if (base + i >= n_text) break;
        if (!SEP) raw_pos[base + i] = pos;                         // characters before text[base+i]
        const uint8_t c = b[i];
        if (SEP) {
            if (c == 0) {                                              // token `tok` starts behind this separator
                ++tok;
                if (sep_ok) { raw_off[tok] = (int32_t)pos; if (base + i == n_text - 1) raw_off[n_tokens] = (int32_t)pos; }      // (a last, empty token)
                continue;
            }
            if (sep_ok && base + i == n_text - 1 && (c & 0xC0) == 0x80) raw_off[n_tokens] = (int32_t)pos;      // (text ends in a continuation byte)
        }
        if ((c & 0xC0) == 0x80) continue;                          // continuation byte
        int cp = -1;
        if (c < 0x80) cp = c;
        else if (c >= 0xC2 && c <= 0xDF && (b[i + 1] & 0xC0) == 0x80 && base + i + 1 < n_text) cp = ((c & 0x1F) << 6) | (b[i + 1] & 0x3F);
        const int byte = (cp >= 0 && cp < 324) ? s_tab[cp] : -1;
        if (byte < 0) atomicMin(err_pos, ((unsigned long long)call << 32) | (SEP ? (uint32_t)tok : (uint32_t)(base + i < 0x7fffffff ? base + i : 0x7ffffffe)));      // earliest call, then earliest position (SEP: token)
        raw[pos] = (uint8_t)(byte < 0 ? 0 : byte);
        ++pos;
        if (SEP && sep_ok && base + i == n_text - 1) raw_off[n_tokens] = (int32_t)pos;      // end of the last token
    }
}

__global__ void token_raw_offsets_kernel(const int32_t* __restrict__ offsets, int64_t n_tokens, int64_t n_text,
                                         const uint32_t* __restrict__ raw_pos, const int32_t* __restrict__ blk_scan,
                                         int n_blocks, int32_t* __restrict__ raw_off) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_tokens) return;
    const int64_t o = offsets[t];
    raw_off[t] = (o >= n_text) ? blk_scan[n_blocks] : (int32_t)raw_pos[o];
}

__global__ void fill_i32_kernel(int32_t* __restrict__ p, int64_t n, int32_t v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

// ---------------------------------------------------------------------------------------
// stage 2: a wave per 64 tokens, one lane per token; the tokens' bytes and their working state live in LDS
// ---------------------------------------------------------------------------------------
// The 64 tokens of a wave are one contiguous range of `raw`: the wave stages it in LDS with coalesced 16-byte loads.
// Every lane then sizes the working state of its token (BPE: symbol list, links and merge queue; Unigram: the Viterbi
// lattice), a wavefront scan over the 64 sizes carves one LDS arena into per-token regions, and the segmentation runs
// on LDS state — the hash tables of the model (2-4 MB: pieces, merges) are probed through L2.  A token whose bytes or
// state do not fit (a wave with > 4 KiB of text, a lane beyond the 24 KiB arena) takes the same code on global memory:
// its region of the `scratch` buffer, sized for the worst case by the host.
struct RetokLds {
    int32_t single_id[256];
    int32_t bf_ids[256];
};

constexpr int RT_TEXT_BYTES = 4096;        // staged text per wave
constexpr int RT_ARENA_WORDS = 6144;       // 24 KiB of per-token state per wave

typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) int32_t lds_i32;
typedef __attribute__((address_space(3))) double lds_f64;
struct GlobalMem { typedef const uint8_t* bytes; typedef int32_t* words; typedef double* doubles; };
struct LdsMem { typedef const lds_u8* bytes; typedef lds_i32* words; typedef lds_f64* doubles; };

// UTF-8 bytes of the printable character standing for raw byte b
__device__ inline int byte_char_utf8(int b, uint8_t out[2]) {
    int cp;
    if ((b >= 33 && b <= 126) || (b >= 161 && b <= 172) || (b >= 174)) cp = b;
    else if (b <= 32) cp = 256 + b;
    else if (b <= 160) cp = 256 + 33 + (b - 127);
    else cp = 256 + 33 + 34;                                        // b == 173
    if (cp < 0x80) { out[0] = (uint8_t)cp; return 1; }
    out[0] = (uint8_t)(0xC0 | (cp >> 6));
    out[1] = (uint8_t)(0x80 | (cp & 0x3F));
    return 2;
}

__device__ inline bool fallback_pair(const RetokLds& L, int b, int32_t out[2], int* n) {
    uint8_t u[2];
    const int nu = byte_char_utf8(b, u);
    for (int k = 0; k < nu; ++k) {
        const int32_t id = L.bf_ids[u[k]];
        if (id < 0) return false;
        out[k] = id;
    }
    *n = nu;
    return true;
}

struct RowWriter {
    int32_t* row;
    int maxlen;
    int n;
    __device__ void push(int32_t id) { if (n < maxlen) row[n] = id; ++n; }
};

// worst-case words of a token's region of the global scratch buffer (the fallback of the LDS arena)
constexpr int SCR_PER_BYTE = 24;
constexpr int SCR_FIXED = 16;

// tokenizers `BPE::merge_word`: the symbols the merges start from (byte fallback / unk / drop).  WRITE = false only counts.
template <bool WRITE, typename M>
__device__ inline int bpe_symbols(const RetokTables& t, const RetokLds& L, typename M::bytes raw, int len, typename M::words c) {
    int n = 0;
    bool unk_pending = false;
    for (int i = 0; i < len; ++i) {
        const int b = raw[i];
        const int32_t id = L.single_id[b];
        if (id >= 0) {
            if (unk_pending) { if (WRITE) c[n] = t.unk_id; ++n; unk_pending = false; }
            if (WRITE) c[n] = id;
            ++n;
            continue;
        }
        if (t.byte_fallback) {
            int32_t fb[2]; int nfb = 0;
            if (fallback_pair(L, b, fb, &nfb)) {            // a pending unk is not flushed first (library behaviour)
                for (int k = 0; k < nfb; ++k) { if (WRITE) c[n] = fb[k]; ++n; }
                continue;
            }
        }
        if (t.unk_id >= 0) {
            if (unk_pending && !t.fuse_unk) { if (WRITE) c[n] = t.unk_id; ++n; }
            unk_pending = true;
        }
    }
    if (unk_pending) { if (WRITE) c[n] = t.unk_id; ++n; }
    return n;
}

// state of a BPE token with n start symbols, in words: c[n] prev[n] next[n] + a queue of (rank, position, new id)
// triples with room for 2n entries — at most n - 1 pairs at the start, one more per merge, at most n - 1 merges
__device__ inline int bpe_state_words(int n) { return 9 * n + 1; }

// A merge-table lookup in two halves, so that several can be in flight at once (r6): merge_probe computes the slot and FETCHES its
// entry, merge_resolve compares and walks on only on a collision.  A lane's lookups used to wait for one another — three dependent
// L2 round trips per merge step (validate the popped pair, then the two new neighbours), one per start pair: the BPE kernel was a
// chain of ~3n round trips per token of n symbols, ~150 us whatever the vocabulary size.
struct MergeHit { bool found; int32_t rank, new_id; };
__device__ inline void merge_probe(const RetokTables& t, int32_t a, int32_t b, uint32_t& slot, MergeEntry& first) {
    slot = merge_slot(a, b, t.merge_mask);
    first = t.merges[slot];
}
__device__ inline MergeHit merge_resolve(const RetokTables& t, int32_t a, int32_t b, uint32_t slot, const MergeEntry& first) {
    if (first.a == -1) return {false, 0, 0};
    if (first.a == a && first.b == b) return {true, first.rank, first.new_id};
    for (slot = (slot + 1) & t.merge_mask;; slot = (slot + 1) & t.merge_mask) {
        const MergeEntry* e = t.merges + slot;
        if (e->a == -1) return {false, 0, 0};
        if (e->a == a && e->b == b) return {true, e->rank, e->new_id};
    }
}

template <typename M>
__device__ inline void bpe_merge(const RetokTables& t, const RetokLds& L, typename M::bytes raw, int len, int n, typename M::words scr, RowWriter& w) {
    typename M::words c = scr;        // symbol ids, -1 = removed
    typename M::words prev = c + n;
    typename M::words next = prev + n;
    typename M::words q = next + n;   // queue triples (rank, pos, new_id)
    bpe_symbols<true, M>(t, L, raw, len, c);
    for (int i = 0; i < n; ++i) { prev[i] = i - 1; next[i] = (i + 1 < n) ? i + 1 : -1; }
    int nq = 0;
    const bool have_merges = t.merge_mask != 0xffffffffu;
    if (have_merges) {
        for (int i0 = 0; i0 + 1 < n; i0 += 4) {          // the start pairs, four lookups in flight
            uint32_t slot[4];
            MergeEntry first[4];
            int32_t a[4], b[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (i0 + k + 1 < n) { a[k] = c[i0 + k]; b[k] = c[i0 + k + 1]; merge_probe(t, a[k], b[k], slot[k], first[k]); }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (i0 + k + 1 >= n) break;
                const MergeHit e = merge_resolve(t, a[k], b[k], slot[k], first[k]);
                if (e.found) { q[3 * nq] = e.rank; q[3 * nq + 1] = i0 + k; q[3 * nq + 2] = e.new_id; ++nq; }
            }
        }
    }
    RETOK_STOP(4);
    while (nq > 0) {                  // Word::merge_all
        int best = 0;
        for (int i = 1; i < nq; ++i)
            if (q[3 * i] < q[3 * best] || (q[3 * i] == q[3 * best] && q[3 * i + 1] < q[3 * best + 1])) best = i;
        const int pos = q[3 * best + 1], new_id = q[3 * best + 2];
        --nq;
        q[3 * best] = q[3 * nq]; q[3 * best + 1] = q[3 * nq + 1]; q[3 * best + 2] = q[3 * nq + 2];
        if (c[pos] < 0 || next[pos] == -1) continue;
        const int r = next[pos], pl = prev[pos], nr = next[r];
        // the popped pair as it stands NOW (an expired entry is one whose pair no longer merges to the same id), and — speculatively,
        // they only count if it is still valid — the two pairs the merged symbol will form with its neighbours: one round trip
        const int32_t ca = c[pos], cb = c[r];
        const int32_t la = pl >= 0 ? c[pl] : 0, rb = nr != -1 ? c[nr] : 0;
        uint32_t s0, s1 = 0, s2 = 0;
        MergeEntry f0, f1{}, f2{};
        merge_probe(t, ca, cb, s0, f0);
        if (pl >= 0) merge_probe(t, la, new_id, s1, f1);
        if (nr != -1) merge_probe(t, new_id, rb, s2, f2);
        const MergeHit e0 = merge_resolve(t, ca, cb, s0, f0);
        if (!e0.found || e0.new_id != new_id) continue;      // expired entry: compared by new id only
        c[pos] = new_id;
        c[r] = -1;
        next[pos] = nr;
        if (nr != -1) prev[nr] = pos;
        if (pl >= 0) {
            const MergeHit e = merge_resolve(t, la, new_id, s1, f1);
            if (e.found) { q[3 * nq] = e.rank; q[3 * nq + 1] = pl; q[3 * nq + 2] = e.new_id; ++nq; }
        }
        if (nr != -1) {
            const MergeHit e = merge_resolve(t, new_id, rb, s2, f2);
            if (e.found) { q[3 * nq] = e.rank; q[3 * nq + 1] = pos; q[3 * nq + 2] = e.new_id; ++nq; }
        }
    }
    RETOK_STOP(5);
    for (int i = 0; i < n; ++i)
        if (c[i] >= 0) w.push(c[i]);
}

// Viterbi lattice of a Unigram token of len bytes: best[len + 1] (double) + bstart / bid / fwd [len + 1]
__device__ inline int unigram_state_words(int len) { return 5 * (len + 1) + 1; }

// the best path of a finished lattice -> ids: backward links reversed, runs of unknown pieces fused, optional byte fallback
template <typename M>
__device__ inline void unigram_emit(const RetokTables& t, const RetokLds& L, typename M::bytes raw, int len, typename M::words bstart,
                                    typename M::words bid, typename M::words fwd, RowWriter& w) {
    for (int e = len; e > 0; e = bstart[e]) fwd[bstart[e]] = e;
    for (int s = 0; s < len;) {
        int e = fwd[s];
        if (bid[e] != -2) { w.push(bid[e]); s = e; continue; }
        int fe = e;                                 // fuse the run of unknown pieces
        while (fe < len && bid[fwd[fe]] == -2) fe = fwd[fe];
        bool ok = t.byte_fallback != 0;
        if (ok) {
            for (int i = s; i < fe && ok; ++i) { int32_t fb[2]; int nfb; ok = fallback_pair(L, raw[i], fb, &nfb); }
        }
        if (ok) {
            for (int i = s; i < fe; ++i) { int32_t fb[2]; int nfb = 0; fallback_pair(L, raw[i], fb, &nfb); for (int k = 0; k < nfb; ++k) w.push(fb[k]); }
        } else {
            w.push(t.unk_id);
        }
        s = fe;
    }
}

// returns false on "unknown token but unk_id is missing"
template <typename M>
__device__ inline bool unigram_token(const RetokTables& t, const RetokLds& L, typename M::bytes raw, int len, typename M::words scr, RowWriter& w) {
    typename M::doubles best = (typename M::doubles)scr;      // [len+1]   (regions start on 8-byte boundaries)
    typename M::words bstart = scr + 2 * (len + 1);           // [len+1]
    typename M::words bid = bstart + len + 1;                 // [len+1]  piece id, -2 = unknown
    typename M::words fwd = bid + len + 1;                    // [len+1]  forward links of the best path
    for (int i = 0; i <= len; ++i) { best[i] = 0.0; bstart[i] = -1; bid[i] = -1; }
    for (int s = 0; s < len; ++s) {
        const double base = best[s];
        bool has_single = false;
        const int emax = (s + t.max_piece_len < len) ? s + t.max_piece_len : len;
        uint64_t h = FNV_OFFSET;
        // The lookups of one start are independent of each other and of the lattice: the first slots of up to four ends are
        // fetched TOGETHER (r5: a lane's probes used to wait for one another — a 32-byte entry of a 16 MiB table is an L2 / MALL
        // miss, ~1-2 us each, and the 250 k-piece XLM-R vocabulary made this kernel 0.26 ms), then resolved in order.
        for (int e0 = s + 1; e0 <= emax; e0 += 4) {
            uint64_t hh[4];
            uint32_t slot[4];
            PieceEntry first[4];
            bool may[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                may[k] = false;
                if (e0 + k <= emax) {
                    h = fnv_step(h, raw[e0 + k - 1]);
                    hh[k] = h;
                    may[k] = piece_maybe(t.piece_bits, t.piece_bits_mask, h);
                    if (may[k]) { slot[k] = piece_slot(h, t.piece_mask); first[k] = t.pieces[slot[k]]; }
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = e0 + k;
                if (e > emax) break;
                if (!may[k]) continue;
                const PieceEntry* p = piece_find_from(t.pieces, t.piece_mask, t.piece_blob, hh[k], raw + s, e - s, slot[k], first[k]);
                if (!p) continue;
                const double cand = p->score + base;
                if (bstart[e] == -1 || cand > best[e]) { best[e] = cand; bstart[e] = s; bid[e] = p->id; }
                if (e == s + 1) has_single = true;
            }
        }
        if (!has_single) {
            if (t.unk_id < 0) return false;
            const double cand = t.unk_score + base;
            const int e = s + 1;
            if (bstart[e] == -1 || cand > best[e]) { best[e] = cand; bstart[e] = s; bid[e] = -2; }
        }
    }
    unigram_emit<M>(t, L, raw, len, bstart, bid, fwd, w);
    return true;
}

// (r6) the same Viterbi walk when every (start, end) lookup of the token has already been made by the workgroup
// (retok_unigram_kernel, phase 1): slot s * W + k of the token's table holds the piece of bytes [s, s + 1 + k) — its id (UG_MISS:
// no such piece) and score.  Same visiting order, same strictly-greater updates, same double additions as unigram_token.
constexpr int32_t UG_MISS = -0x7fffffff - 1;
template <typename M>
__device__ inline bool unigram_token_table(const RetokTables& t, const RetokLds& L, typename M::bytes raw, int len, typename M::words scr, RowWriter& w,
                                           const lds_i32* pid, const lds_f64* pscore, int W) {
    typename M::doubles best = (typename M::doubles)scr;
    typename M::words bstart = scr + 2 * (len + 1);
    typename M::words bid = bstart + len + 1;
    typename M::words fwd = bid + len + 1;
    for (int i = 0; i <= len; ++i) { best[i] = 0.0; bstart[i] = -1; bid[i] = -1; }
    for (int s = 0; s < len; ++s) {
        const double base = best[s];
        bool has_single = false;
        const int kmax = (len - s < W) ? len - s : W;
        // four ends of the start at a time: their table slots and lattice cells are read TOGETHER (distinct ends: no two of the
        // four updates touch the same cell), then compared and written — the walk is a chain of LDS latencies otherwise
        for (int k0 = 0; k0 < kmax; k0 += 4) {
            int32_t id[4];
            double sc[4], be[4];
            int bs[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool in = k0 + k < kmax;
                const int slot = s * W + (in ? k0 + k : k0), e = s + 1 + (in ? k0 + k : k0);
                id[k] = in ? pid[slot] : UG_MISS;
                sc[k] = pscore[slot];
                be[k] = best[e];
                bs[k] = bstart[e];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (id[k] == UG_MISS) continue;
                const int e = s + 1 + k0 + k;
                const double cand = sc[k] + base;
                if (bs[k] == -1 || cand > be[k]) { best[e] = cand; bstart[e] = s; bid[e] = id[k]; }
                if (k0 + k == 0) has_single = true;
            }
        }
        if (!has_single) {
            if (t.unk_id < 0) return false;
            const double cand = t.unk_score + base;
            const int e = s + 1;
            if (bstart[e] == -1 || cand > best[e]) { best[e] = cand; bstart[e] = s; bid[e] = -2; }
        }
    }
#ifdef ZETT_RETOK_DEBUG
    if (t.debug_stop == 4) return true;
#endif
    unigram_emit<M>(t, L, raw, len, bstart, bid, fwd, w);
    return true;
}

// tokenizers `WordPiece::tokenize`.  Returns false on "the word is [UNK] but [UNK] is not in the vocabulary".  From each
// start the library tries the longest substring first and shortens it a character at a time; walking the ends upwards
// with a running hash and keeping the LAST hit finds the same piece.  One byte-level character is one raw byte.
template <typename M>
__device__ inline bool wordpiece_token(const RetokTables& t, typename M::bytes raw, int len, RowWriter& w, int32_t pad_id) {
    if (len > t.max_word_chars) {
        if (t.unk_id < 0) return false;
        w.push(t.unk_id);
        return true;
    }
    const int n0 = w.n;
    for (int s = 0; s < len;) {
        const int where = s == 0 ? 0 : 1;
        uint64_t h = where ? FNV_OFFSET_CONT : FNV_OFFSET;
        const int emax = (s + t.max_piece_len < len) ? s + t.max_piece_len : len;
        int best_e = -1, best_id = -1;
        for (int e = s + 1; e <= emax; ++e) {
            h = fnv_step(h, raw[e - 1]);
            const PieceEntry* p = piece_find_where(t.pieces, t.piece_mask, t.piece_blob, h, raw + s, e - s, where);
            if (p) { best_e = e; best_id = p->id; }
        }
        if (best_e < 0) {                                   // is_bad: the pieces found so far are dropped, the word is [UNK]
            if (t.unk_id < 0) return false;
            for (int i = n0; i < w.n && i < w.maxlen; ++i) w.row[i] = pad_id;
            w.n = n0;
            w.push(t.unk_id);
            return true;
        }
        w.push(best_id);
        s = best_e;
    }
    return true;
}

// special-token lookup (zett/utils.py:671-673) and, with ignore_merges, the whole-token lookup of BPE::tokenize: -1 = no hit
template <typename M>
__device__ inline int whole_token_id(const PieceEntry* tab, uint32_t mask, const uint8_t* blob, typename M::bytes s, int len) {
    uint64_t h = FNV_OFFSET;
    for (int i = 0; i < len; ++i) h = fnv_step(h, s[i]);
    const PieceEntry* e = piece_find(tab, mask, blob, h, s, len);
    return e ? e->id : -1;
}

// one token: `n_sym` = its BPE start symbols (0 for the other kinds), `scr` = its state region
template <typename M>
__device__ inline bool segment_token(const RetokTables& t, const RetokLds& L, typename M::bytes s, int len, int n_sym, typename M::words scr,
                                     RowWriter& w, int32_t pad_id) {
    if (t.kind == ZETT_RETOK_BPE) { bpe_merge<M>(t, L, s, len, n_sym, scr, w); return true; }
    if (t.kind == ZETT_RETOK_UNIGRAM) return unigram_token<M>(t, L, s, len, scr, w);
    return wordpiece_token<M>(t, s, len, w, pad_id);
}

__global__ __launch_bounds__(64) void retok_tokens_kernel(RetokTables t, const uint8_t* __restrict__ raw,
                                                          const int32_t* __restrict__ raw_off, int64_t n_tokens,
                                                          int maxlen, int32_t pad_id, int32_t* __restrict__ out, int32_t* __restrict__ scratch,
                                                          unsigned long long* __restrict__ n_truncated,
                                                          unsigned long long* __restrict__ err_unk, uint32_t call) {
    __shared__ RetokLds L;
    __shared__ __attribute__((aligned(16))) uint8_t s_text[RT_TEXT_BYTES + 16];
    __shared__ __attribute__((aligned(8))) int32_t s_arena[RT_ARENA_WORDS];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) { L.single_id[i] = t.single_id[i]; L.bf_ids[i] = t.bf_ids[i]; }
    const int64_t tok0 = (int64_t)blockIdx.x * 64;
    const int64_t tok = tok0 + lane;
    const bool live = tok < n_tokens;
    const int o0 = live ? raw_off[tok] : 0;
    const int len = live ? raw_off[tok + 1] - o0 : 0;
    // the wave's text: bytes [w_lo, w_hi) of `raw`, staged from the 16-byte boundary below w_lo (the buffer has 16 bytes of slack)
    const int w_lo = raw_off[tok0];
    const int w_hi = raw_off[tok0 + 64 < n_tokens ? tok0 + 64 : n_tokens];
    const int mis = w_lo & 15;
    const bool text_lds = (w_hi - w_lo) + mis <= RT_TEXT_BYTES;
    if (text_lds)
        for (int i = lane * 16; i < (w_hi - w_lo) + mis; i += 64 * 16) *(uint4*)(s_text + i) = *(const uint4*)(raw + (w_lo - mis) + i);
    __syncthreads();
    RETOK_STOP(1);
    const lds_u8* sl = (const lds_u8*)s_text + mis + (o0 - w_lo);
    const uint8_t* sg = raw + o0;
    RowWriter w{out + tok * maxlen, maxlen, 0};
    bool todo = live && len > 0;
    if (todo && t.special_mask != 0xffffffffu) {              // zett/utils.py:671-673
        const int id = text_lds ? whole_token_id<LdsMem>(t.specials, t.special_mask, t.special_blob, sl, len)
                                : whole_token_id<GlobalMem>(t.specials, t.special_mask, t.special_blob, sg, len);
        if (id >= 0) { w.row[0] = id; todo = false; }
    }
    if (todo && t.kind == ZETT_RETOK_BPE && t.ignore_merges) {
        const int id = text_lds ? whole_token_id<LdsMem>(t.pieces, t.piece_mask, t.piece_blob, sl, len)
                                : whole_token_id<GlobalMem>(t.pieces, t.piece_mask, t.piece_blob, sg, len);
        if (id >= 0) { w.push(id); todo = false; }
    }
    RETOK_STOP(2);
    // state of the token, and its place in the wave's arena: an exclusive wavefront scan over the 64 sizes
    int n_sym = 0, need = 0;
    if (todo) {
        if (t.kind == ZETT_RETOK_BPE) {
            n_sym = text_lds ? bpe_symbols<false, LdsMem>(t, L, sl, len, (lds_i32*)nullptr) : bpe_symbols<false, GlobalMem>(t, L, sg, len, (int32_t*)nullptr);
            need = bpe_state_words(n_sym);
        } else if (t.kind == ZETT_RETOK_UNIGRAM) {
            need = unigram_state_words(len);
        }
        need = (need + 1) & ~1;                                // regions start on 8-byte boundaries
    }
    int inc = need;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(inc, off, 64);
        if (lane >= off) inc += v;
    }
    const int a0 = inc - need;
    RETOK_STOP(3);
    if (todo) {
        bool ok;
        if (text_lds && a0 + need <= RT_ARENA_WORDS) {
            ok = segment_token<LdsMem>(t, L, sl, len, n_sym, (lds_i32*)s_arena + a0, w, pad_id);
        } else {
            int32_t* scr = scratch + ((int64_t)o0 * SCR_PER_BYTE + tok * SCR_FIXED);
            ok = segment_token<GlobalMem>(t, L, sg, len, n_sym, scr, w, pad_id);
        }
        if (!ok) {
            atomicMin(err_unk, ((unsigned long long)call << 32) | (uint32_t)(tok < 0x7ffffffe ? tok : 0x7ffffffe));      // earliest call, then earliest token
            return;
        }
        if (w.n > maxlen) atomicAdd(n_truncated, 1ull);      // zett/utils.py:683-685
    }
}

// ---------------------------------------------------------------------------------------
// stage 2 for Unigram models (r6): a WORKGROUP per 64 tokens — the piece lookups by all four waves, the Viterbi walk by lane
// ---------------------------------------------------------------------------------------
// The one-lane-per-token kernel above runs a Unigram token as len starts x up to max_piece_len ends of DEPENDENT table probes
// (a 32-byte entry of a 16 MiB table: an L2 / MALL miss, ~1-2 us), and a wave's 64 lanes walk tokens of different lengths in
// lockstep: 0.25 ms for the 50 k tokens of XLM-R -> GPT-2 whatever the occupancy (NOTEBOOK R5.5).  But the lookups do not depend
// on the lattice at all — only the score accumulation does.  So:
//   phase 0  (wave 0, a lane per token) special-token lookup, sizes: the token's (start, end) table has len x W slots,
//            W = min(len, max_piece_len); a wavefront scan places the tables of the 64 tokens in one LDS array
//   phase 1  (256 threads) the slots of all tokens as ONE flat work list: thread -> (token, start, end) by a binary search over
//            the 65 prefix sums, FNV hash of the substring from the staged text, first table entries of four slots fetched
//            together, then resolved; (id, score) or a miss into the slot.  Every probe of the workgroup is in flight at once:
//            ~9 slots per thread instead of ~150 dependent round trips per lane
//   phase 2  (wave 0, a lane per token) unigram_token_table: the same walk on LDS reads, then the same emission
// Tokens whose tables do not fit the 4 096 slots together are processed in rounds; a token that alone exceeds them, and a
// workgroup whose text does not fit its LDS stage, take the per-lane code.  Integer results, same visiting order: identical ids.
constexpr int UG_THREADS = 256;
constexpr int UG_TOKENS = 64;
constexpr int UG_PCAP = 4096;

__global__ __launch_bounds__(UG_THREADS) void retok_unigram_kernel(RetokTables t, const uint8_t* __restrict__ raw,
                                                                   const int32_t* __restrict__ raw_off, int64_t n_tokens,
                                                                   int maxlen, int32_t pad_id, int32_t* __restrict__ out, int32_t* __restrict__ scratch,
                                                                   unsigned long long* __restrict__ n_truncated,
                                                                   unsigned long long* __restrict__ err_unk, uint32_t call) {
    __shared__ RetokLds L;
    __shared__ __attribute__((aligned(16))) uint8_t s_text[RT_TEXT_BYTES + 16];
    __shared__ __attribute__((aligned(8))) int32_t s_arena[RT_ARENA_WORDS];
    __shared__ __attribute__((aligned(8))) double s_pscore[UG_PCAP];
    __shared__ int32_t s_pid[UG_PCAP];
    __shared__ int s_base[UG_TOKENS + 1];          // first slot of every token's table (0 slots: nothing to segment)
    __shared__ int s_len[UG_TOKENS], s_toff[UG_TOKENS];
    const int tid = threadIdx.x, lane = tid & 63;
    const bool wave0 = tid < 64;
    for (int i = tid; i < 256; i += UG_THREADS) { L.single_id[i] = t.single_id[i]; L.bf_ids[i] = t.bf_ids[i]; }
    const int64_t tok0 = (int64_t)blockIdx.x * UG_TOKENS;
    const int64_t tok = tok0 + lane;
    const bool live = wave0 && tok < n_tokens;
    const int o0 = live ? raw_off[tok] : 0;
    const int len = live ? raw_off[tok + 1] - o0 : 0;
    const int w_lo = raw_off[tok0];
    const int w_hi = raw_off[tok0 + UG_TOKENS < n_tokens ? tok0 + UG_TOKENS : n_tokens];
    const int mis = w_lo & 15;
    const bool text_lds = (w_hi - w_lo) + mis <= RT_TEXT_BYTES;          // (uniform)
    if (text_lds)
        for (int i = tid * 16; i < (w_hi - w_lo) + mis; i += UG_THREADS * 16) *(uint4*)(s_text + i) = *(const uint4*)(raw + (w_lo - mis) + i);
    __syncthreads();
    RETOK_STOP(1);
    const lds_u8* sl = (const lds_u8*)s_text + mis + (o0 - w_lo);
    const uint8_t* sg = raw + o0;
    RowWriter w{out + (live ? tok : 0) * maxlen, maxlen, 0};
    bool todo = live && len > 0;
    if (todo && t.special_mask != 0xffffffffu) {              // zett/utils.py:671-673
        const int id = text_lds ? whole_token_id<LdsMem>(t.specials, t.special_mask, t.special_blob, sl, len)
                                : whole_token_id<GlobalMem>(t.specials, t.special_mask, t.special_blob, sg, len);
        if (id >= 0) { w.row[0] = id; todo = false; }
    }
    int need = todo ? ((unigram_state_words(len) + 1) & ~1) : 0;
    const int W_mine = len < t.max_piece_len ? len : t.max_piece_len;
    int slots = (todo && text_lds) ? len * W_mine : 0;
    int a0 = 0;
    if (wave0) {
        int inc = need, inc_s = slots;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(inc, off, 64), vs = __shfl_up(inc_s, off, 64);
            if (lane >= off) { inc += v; inc_s += vs; }
        }
        a0 = inc - need;
        s_base[lane + 1] = inc_s;
        if (lane == 0) s_base[0] = 0;
        s_len[lane] = todo ? len : 0;
        s_toff[lane] = mis + (o0 - w_lo);
    }
    __syncthreads();
    RETOK_STOP(2);
    bool ok = true;
    auto finish = [&]() {
        if (!todo) return;
        if (!ok) { atomicMin(err_unk, ((unsigned long long)call << 32) | (uint32_t)(tok < 0x7ffffffe ? tok : 0x7ffffffe)); return; }
        if (w.n > maxlen) atomicAdd(n_truncated, 1ull);       // zett/utils.py:683-685
    };
    if (!text_lds) {          // (uniform) more than 4 KiB of text in 64 tokens: the per-lane code on global memory
        if (todo) {
            int32_t* scr = scratch + ((int64_t)o0 * SCR_PER_BYTE + tok * SCR_FIXED);
            ok = unigram_token<GlobalMem>(t, L, sg, len, scr, w);
        }
        finish();
        return;
    }
    for (int tb = 0; tb < UG_TOKENS;) {          // rounds of tokens [tb, te) whose tables fit together (all of it uniform: LDS values only)
        int te = tb;
        while (te < UG_TOKENS && s_base[te + 1] - s_base[tb] <= UG_PCAP) ++te;
        const bool solo = te == tb;               // token tb alone exceeds the table: the per-lane code
        if (solo) te = tb + 1;
        const int b0 = s_base[tb], total = solo ? 0 : s_base[te] - b0;
        // ---- phase 1: every (token, start, end) slot of the round, four per thread at a time
        for (int i = tid; i < total; i += UG_THREADS * 4) {
            uint64_t hh[4];
            uint32_t slot[4];
            PieceEntry first[4];
            int sub_off[4], sub_len[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = i + k * UG_THREADS;
                sub_len[k] = 0;
                if (idx >= total) continue;
                int lo = tb, hi = te - 1;              // the last token whose table starts at or before b0 + idx
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (s_base[mid] - b0 <= idx) lo = mid; else hi = mid - 1;
                }
                const int tl = s_len[lo];
                const int Wt = tl < t.max_piece_len ? tl : t.max_piece_len;
                const int local = idx - (s_base[lo] - b0);
                const int st = local / Wt, e = st + 1 + local % Wt;
                if (e > tl) { s_pid[idx] = UG_MISS; continue; }          // (the tail of the table's last rows: no such substring)
                sub_off[k] = s_toff[lo] + st;
                sub_len[k] = e - st;
                uint64_t h = FNV_OFFSET;
                for (int b = 0; b < sub_len[k]; ++b) h = fnv_step(h, s_text[sub_off[k] + b]);
                hh[k] = h;
                if (!piece_maybe(t.piece_bits, t.piece_bits_mask, h)) { s_pid[idx] = UG_MISS; sub_len[k] = 0; continue; }      // a certain miss, answered from L2
                slot[k] = piece_slot(h, t.piece_mask);
                first[k] = t.pieces[slot[k]];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (sub_len[k] == 0) continue;
                const int idx = i + k * UG_THREADS;
                const PieceEntry* pe = piece_find_from(t.pieces, t.piece_mask, t.piece_blob, hh[k], (const lds_u8*)s_text + sub_off[k], sub_len[k], slot[k], first[k]);
                if (pe) { s_pid[idx] = pe->id; s_pscore[idx] = pe->score; }
                else s_pid[idx] = UG_MISS;
            }
        }
        __syncthreads();
        RETOK_STOP(3);
        // ---- phase 2: the walk, a lane per token
        if (wave0 && todo && lane >= tb && lane < te) {
            const bool arena = a0 + need <= RT_ARENA_WORDS;
            int32_t* scr = scratch + ((int64_t)o0 * SCR_PER_BYTE + tok * SCR_FIXED);
            if (solo) {
                ok = arena ? unigram_token<LdsMem>(t, L, sl, len, (lds_i32*)s_arena + a0, w) : unigram_token<GlobalMem>(t, L, sg, len, scr, w);
            } else {
                const lds_i32* pid = (const lds_i32*)s_pid + (s_base[lane] - b0);
                const lds_f64* psc = (const lds_f64*)s_pscore + (s_base[lane] - b0);
                ok = arena ? unigram_token_table<LdsMem>(t, L, sl, len, (lds_i32*)s_arena + a0, w, pid, psc, W_mine)
                           : unigram_token_table<GlobalMem>(t, L, sg, len, scr, w, pid, psc, W_mine);
            }
            finish();
        }
        __syncthreads();
        tb = te;
    }
}

}  // namespace zett

// =========================================================================================
// host side
// =========================================================================================
struct zett_retok {
    int device = 0;
    zett::RetokTables t{};
    std::vector<void*> owned;
    zett::DevBuf raw, raw_pos, raw_off, blk, scratch, misc;
    int32_t* host_pinned = nullptr;      // [0..1] text length (sync entry point); [4..9] the three 64-bit result words; [12..17] their initial values
    hipEvent_t done = nullptr;           // behind the last enqueued call's result copy
    uint32_t calls = 0;                  // calls enqueued since the last result query (ordinal of the next call)
    struct Call { const int32_t* offsets; int64_t n_tokens; };
    std::vector<Call> recent;            // the calls since the last result query (for the token of a KeyError)
    bool words_ready = false;            // the device result words hold their initial values
    int unigram_wg = 1;                  // Unigram models: 1 = the workgroup-per-64-tokens kernel for calls of up to 32 768 tokens, 2 = always, 0 = never (the lane-per-token kernel)
};

namespace zett {

// Open addressing with linear probing, load factor 1/16 .. 1/8 (r6; was 1/4 .. 1/2).  A wave waits for the LONGEST probe chain among
// its 64 lanes' lookups, every probe a dependent L2 / MALL round trip, and the clusters of linear probing at half load have a heavy
// tail: the phase timing of round 6 (tools/retok_ab.py on a -DZETT_RETOK_DEBUG build) put 48 of a Unigram workgroup's 87 us into
// its lookups and 70 of the BPE kernel's 135 into the merge loop.  Memory is not the constraint on a 288 GB part (XLM-R's 250 k
// pieces: 64 MB; Llama-3's 128 k merges: 16 MB).
inline uint32_t pow2_capacity(size_t n) {
    uint32_t c = 16;
    while (c < 8 * n + 2) c <<= 1;
    return c;
}

struct HostPieceTable {
    std::vector<PieceEntry> slots;
    std::vector<uint8_t> blob;
    std::vector<uint32_t> bits;          // piece_maybe's bitmap (word-initial WordPiece / all other pieces alike: keyed by the hash)
    uint32_t bits_mask = 0;
    uint32_t mask = 0xffffffffu;
    int max_len = 0;
};

inline void build_piece_table(const uint8_t* bytes, const int32_t* offsets, const int32_t* ids, const double* scores, int n,
                              HostPieceTable& out, int32_t* single_id /* nullable */, const uint8_t* where = nullptr /* WordPiece: 1 = continuing piece */) {
    // duplicates: the LAST listed piece wins (tokenizers inserts into a HashMap in listing order)
    std::unordered_map<std::string, int> last;
    last.reserve((size_t)n * 2);
    for (int i = 0; i < n; ++i) {
        const int len = offsets[i + 1] - offsets[i];
        if (len <= 0) continue;
        std::string key((const char*)bytes + offsets[i], (size_t)len);
        if (where) key.push_back(where[i] ? '\1' : '\0');        // the key space is part of the key
        last[key] = i;
    }
    if (last.empty()) return;
    const uint32_t cap = pow2_capacity(last.size());
    out.mask = cap - 1;
    out.slots.assign(cap, PieceEntry{0, 0, 0, 0, 0, 0.0});
    out.bits_mask = cap * 4 - 1;         // 32 .. 64 bits per piece
    out.bits.assign((size_t)cap * 4 / 32, 0u);
    for (auto& kv : last) {
        const int i = kv.second;
        const int len = (int)kv.first.size() - (where ? 1 : 0);
        const int wh = where ? (where[i] ? 1 : 0) : 0;
        uint64_t h = wh ? FNV_OFFSET_CONT : FNV_OFFSET;
        for (int k = 0; k < len; ++k) h = fnv_step(h, (uint8_t)kv.first[k]);
        PieceEntry e{h, (int32_t)out.blob.size(), len, ids[i], wh, scores ? scores[i] : 0.0};
        out.blob.insert(out.blob.end(), kv.first.begin(), kv.first.begin() + len);
        uint32_t slot = piece_slot(h, out.mask);
        while (out.slots[slot].len != 0) slot = (slot + 1) & out.mask;
        out.slots[slot] = e;
        { const uint32_t b = piece_bit(h, out.bits_mask); out.bits[b >> 5] |= 1u << (b & 31); }
        if (len > out.max_len) out.max_len = len;
        if (single_id && len == 1 && !wh) single_id[(uint8_t)kv.first[0]] = ids[i];
    }
    out.blob.resize(out.blob.size() + 16, 0);          // slack for piece_bytes_equal's whole-word reads
}

template <typename U>
inline int upload(zett_retok* r, const std::vector<U>& v, const U** dev) {
    *dev = nullptr;
    if (v.empty()) return 0;
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, v.size() * sizeof(U)));
    r->owned.push_back(p);
    HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(U), hipMemcpyHostToDevice));
    *dev = (const U*)p;
    return 0;
}

}  // namespace zett

extern "C" {

int zett_retok_create(const zett_retok_model* m, int device, zett_retok** out) {
    using namespace zett;
    if (!m || !out) return fail(ZETT_E_INVALID, "null argument");
    if (m->kind != ZETT_RETOK_BPE && m->kind != ZETT_RETOK_UNIGRAM && m->kind != ZETT_RETOK_WORDPIECE) return fail(ZETT_E_NOT_IMPLEMENTED, "hn tokenizer model kind %d", m->kind);
    if (m->kind == ZETT_RETOK_WORDPIECE && m->max_input_chars_per_word < 0) return fail(ZETT_E_INVALID, "max_input_chars_per_word must be >= 0");
    if (m->n_pieces < 0 || m->n_merges < 0 || m->n_special < 0) return fail(ZETT_E_INVALID, "negative count");
    if (m->n_pieces && (!m->piece_bytes || !m->piece_offsets || !m->piece_ids)) return fail(ZETT_E_INVALID, "piece arrays missing");
    if (m->kind == ZETT_RETOK_UNIGRAM && m->n_pieces && !m->piece_scores) return fail(ZETT_E_INVALID, "Unigram needs piece_scores");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(ZETT_E_INVALID, "device %d out of range", device);
    ZETT_ON_DEVICE(device);
    auto* r = new zett_retok();
    r->device = device;
    std::vector<int32_t> single(256, -1), bf(256, -1);
    if (m->byte_fallback && m->byte_fallback_ids) bf.assign(m->byte_fallback_ids, m->byte_fallback_ids + 256);
    HostPieceTable pt, st;
    build_piece_table(m->piece_bytes, m->piece_offsets, m->piece_ids, m->piece_scores, m->n_pieces, pt, single.data(),
                      m->kind == ZETT_RETOK_WORDPIECE ? m->piece_continuing : nullptr);
    build_piece_table(m->special_bytes, m->special_offsets, m->special_ids, nullptr, m->n_special, st, nullptr);
    std::vector<MergeEntry> mt;
    uint32_t mmask = 0xffffffffu;
    if (m->n_merges > 0) {
        const uint32_t cap = pow2_capacity((size_t)m->n_merges);
        mmask = cap - 1;
        mt.assign(cap, MergeEntry{-1, -1, 0, 0});
        for (int i = 0; i < m->n_merges; ++i) {   // a pair listed twice: the later entry wins (HashMap collect)
            const int32_t a = m->merges[3 * i], b = m->merges[3 * i + 1], nid = m->merges[3 * i + 2];
            if (a < 0 || b < 0) { delete r; return fail(ZETT_E_INVALID, "merge %d has a negative id", i); }
            uint32_t slot = merge_slot(a, b, mmask);
            while (mt[slot].a != -1 && !(mt[slot].a == a && mt[slot].b == b)) slot = (slot + 1) & mmask;
            mt[slot] = MergeEntry{a, b, i, nid};
        }
    }
    std::vector<int16_t> cp(324, -1);           // GPT-2 bytes_to_unicode, inverted
    {
        int extra = 0;
        for (int b = 0; b < 256; ++b) {
            const bool keep = (b >= 33 && b <= 126) || (b >= 161 && b <= 172) || (b >= 174);
            cp[keep ? b : 256 + extra++] = (int16_t)b;
        }
    }
    RetokTables& t = r->t;
    int rc = 0;
    const PieceEntry* dp = nullptr; const uint8_t* db = nullptr; const MergeEntry* dm = nullptr;
    const int32_t* di = nullptr; const int16_t* dc = nullptr;
    if ((rc = upload(r, pt.slots, &dp))) { zett_retok_destroy(r); return rc; } t.pieces = dp;
    if ((rc = upload(r, pt.blob, &db))) { zett_retok_destroy(r); return rc; } t.piece_blob = db;
    t.piece_mask = pt.mask;
    { const uint32_t* dbits = nullptr;
      if ((rc = upload(r, pt.bits, &dbits))) { zett_retok_destroy(r); return rc; }
      t.piece_bits = dbits; t.piece_bits_mask = pt.bits_mask; }
    if (!dp) {   // empty vocabulary: one empty slot so lookups terminate
        std::vector<PieceEntry> one(16, PieceEntry{0, 0, 0, 0, 0, 0.0});
        if ((rc = upload(r, one, &dp))) { zett_retok_destroy(r); return rc; }
        t.pieces = dp; t.piece_mask = 15;
    }
    if ((rc = upload(r, st.slots, &dp))) { zett_retok_destroy(r); return rc; } t.specials = dp;
    if ((rc = upload(r, st.blob, &db))) { zett_retok_destroy(r); return rc; } t.special_blob = db;
    t.special_mask = st.mask;
    if ((rc = upload(r, mt, &dm))) { zett_retok_destroy(r); return rc; } t.merges = dm; t.merge_mask = mmask;
    if ((rc = upload(r, single, &di))) { zett_retok_destroy(r); return rc; } t.single_id = di;
    if ((rc = upload(r, bf, &di))) { zett_retok_destroy(r); return rc; } t.bf_ids = di;
    if ((rc = upload(r, cp, &dc))) { zett_retok_destroy(r); return rc; } t.cp_to_byte = dc;
    t.kind = m->kind; t.unk_id = m->unk_id; t.fuse_unk = m->fuse_unk; t.byte_fallback = m->byte_fallback;
    t.ignore_merges = m->ignore_merges; t.max_piece_len = pt.max_len; t.max_word_chars = m->max_input_chars_per_word;
#ifdef ZETT_RETOK_DEBUG
    { const char* e = getenv("ZETT_RETOK_STOP"); t.debug_stop = e ? atoi(e) : 0; }
#endif
    t.unk_score = m->unigram_min_score - 10.0;   // tokenizers kUnkPenalty
    HIP_TRY(hipHostMalloc((void**)&r->host_pinned, 128, hipHostMallocDefault));
    HIP_TRY(hipEventCreateWithFlags(&r->done, hipEventDisableTiming));
    *out = r;
    return 0;
}

int zett_retok_set_option(zett_retok* r, const char* key, int64_t value) {
    using namespace zett;
    if (!r || !key) return fail(ZETT_E_INVALID, "null argument");
    if (!strcmp(key, "unigram_workgroup")) {
        if (value < 0 || value > 2) return fail(ZETT_E_INVALID, "unigram_workgroup: 0 (lane kernel), 1 (by size), 2 (workgroup kernel)");
        r->unigram_wg = (int)value;
        return 0;
    }
    return fail(ZETT_E_INVALID, "unknown retokenizer option '%s'", key);
}

int zett_retok_destroy(zett_retok* r) {
    if (!r) return 0;
    ::zett::DeviceScope _scope(r->device);
    for (void* p : r->owned) (void)hipFree(p);
    for (zett::DevBuf* b : {&r->raw, &r->raw_pos, &r->raw_off, &r->blk, &r->scratch, &r->misc}) b->release();
    if (r->host_pinned) (void)hipHostFree(r->host_pinned);
    if (r->done) (void)hipEventDestroy(r->done);
    delete r;
    return 0;
}

// result words on the device (r->misc): [0] err_pos = (call << 32 | text position) of the first character outside the byte
// table, [1] err_unk = (call << 32 | token) of the first token that needed a missing unk id, [2] tokens cut to maxlen —
// accumulated over the calls since the last result query
static int retok_reset_words(zett_retok* r, hipStream_t st) {
    unsigned long long* init = (unsigned long long*)(r->host_pinned + 12);
    init[0] = ~0ull; init[1] = ~0ull; init[2] = 0ull;
    if (int rc = r->misc.reserve(64)) return rc;
    HIP_TRY(hipMemcpyAsync(r->misc.p, init, 24, hipMemcpyHostToDevice, st));
    r->words_ready = true;
    r->calls = 0;
    r->recent.clear();
    return 0;
}

int zett_retokenize_async(zett_retok* r, const uint8_t* token_chars, const int32_t* offsets, int64_t n_tokens, int64_t n_text,
                          int32_t maxlen, int32_t pad_id, int32_t* out, void* stream) {
    using namespace zett;
    if (!r) return fail(ZETT_E_INVALID, "null argument");
    if (n_tokens < 0 || maxlen < 1 || n_text < 0 || n_text >= (int64_t)0x7fffffff) return fail(ZETT_E_INVALID, "bad shape");
    if (n_tokens == 0) return 0;
    if (!out) return fail(ZETT_E_INVALID, "null argument");
    if (n_text > 0 && !token_chars) return fail(ZETT_E_INVALID, "token_chars is null");
    const bool sep = offsets == nullptr;      // NUL-separated text: the token boundaries are found on the device
    if (sep && n_text < n_tokens - 1) return fail(ZETT_E_INVALID, "NUL-separated text of %lld tokens needs at least %lld bytes", (long long)n_tokens, (long long)(n_tokens - 1));
    ZETT_ON_DEVICE(r->device);
    hipStream_t st = (hipStream_t)stream;
    if (!r->words_ready) { if (int rc = retok_reset_words(r, st)) return rc; }
    if (r->calls >= (1u << 20)) return fail(ZETT_E_STATE, "2^20 asynchronous calls without zett_retok_result: collect the results first");
    const int n_blocks = (int)((n_text + CH_PER_BLOCK - 1) / CH_PER_BLOCK);
    if (int rc = r->raw.reserve((size_t)n_text + 16)) return rc;
    if (int rc = r->raw_pos.reserve(((size_t)n_text + 1) * 4)) return rc;
    if (int rc = r->raw_off.reserve(((size_t)n_tokens + 1) * 4)) return rc;
    if (int rc = r->blk.reserve(((size_t)n_blocks + 2) * 4 * 4)) return rc;
    if (int rc = r->scratch.reserve(((size_t)n_text * SCR_PER_BYTE + (size_t)n_tokens * SCR_FIXED + 64) * 4)) return rc;
    int32_t* blk_count = r->blk.as<int32_t>();
    int32_t* blk_scan = blk_count + n_blocks + 1;
    int32_t* sep_count = blk_scan + n_blocks + 2;          // (separator mode: the same pair for the NUL counts)
    int32_t* sep_scan = sep_count + n_blocks + 1;
    unsigned long long* words = r->misc.as<unsigned long long>();
    const uint32_t call = r->calls++;
    r->recent.push_back({offsets, n_tokens});          // (the caller keeps `offsets` alive until zett_retok_result: include/zett_hip.h)
    // a failed enqueue takes its call back: zett_retok_result must not wait on a `done` event that was never recorded for it,
    // and the words (which kernels of this call may already have touched) start afresh with the next call
    auto enqueue = [&]() -> int {
        hipLaunchKernelGGL(fill_i32_kernel, dim3(1024), dim3(256), 0, st, out, n_tokens * (int64_t)maxlen, pad_id);   // :662-666
        if (sep) HIP_TRY(hipMemsetAsync(r->raw_off.p, 0, ((size_t)n_tokens + 1) * 4, st));
        if (n_blocks > 0 && sep) {
            hipLaunchKernelGGL((chars_to_bytes_kernel<0, true>), dim3(n_blocks), dim3(256), 0, st, token_chars, n_text, r->t.cp_to_byte,
                               blk_count, (uint8_t*)nullptr, (uint32_t*)nullptr, (unsigned long long*)nullptr, call, sep_count, (int32_t*)nullptr, n_tokens);
            hipLaunchKernelGGL(exclusive_scan_kernel, dim3(1), dim3(1024), 0, st, blk_count, blk_scan, (int64_t)n_blocks);
            hipLaunchKernelGGL(exclusive_scan_kernel, dim3(1), dim3(1024), 0, st, sep_count, sep_scan, (int64_t)n_blocks);
            hipLaunchKernelGGL((chars_to_bytes_kernel<1, true>), dim3(n_blocks), dim3(256), 0, st, token_chars, n_text, r->t.cp_to_byte,
                               blk_scan, r->raw.as<uint8_t>(), (uint32_t*)nullptr, words, call, sep_scan, r->raw_off.as<int32_t>(), n_tokens);
        } else if (n_blocks > 0) {
            hipLaunchKernelGGL((chars_to_bytes_kernel<0, false>), dim3(n_blocks), dim3(256), 0, st, token_chars, n_text, r->t.cp_to_byte,
                               blk_count, (uint8_t*)nullptr, (uint32_t*)nullptr, (unsigned long long*)nullptr, call);
            hipLaunchKernelGGL(exclusive_scan_kernel, dim3(1), dim3(1024), 0, st, blk_count, blk_scan, (int64_t)n_blocks);
            hipLaunchKernelGGL((chars_to_bytes_kernel<1, false>), dim3(n_blocks), dim3(256), 0, st, token_chars, n_text, r->t.cp_to_byte,
                               blk_scan, r->raw.as<uint8_t>(), r->raw_pos.as<uint32_t>(), words, call);
        } else {
            HIP_TRY(hipMemsetAsync(blk_scan, 0, 8, st));
        }
        if (!sep)
            hipLaunchKernelGGL(token_raw_offsets_kernel, dim3((unsigned)((n_tokens + 1 + 255) / 256)), dim3(256), 0, st, offsets, n_tokens,
                               n_text, r->raw_pos.as<uint32_t>(), blk_scan, n_blocks, r->raw_off.as<int32_t>());
        // Unigram: the workgroup kernel while its workgroups (two per CU) fit the chip in one round — a rank's shard, a CLI batch —,
        // the lane kernel beyond (five single-wave workgroups per CU, all resident): 4 096 tokens 74 vs 90 us, 50 350 tokens 125 vs 103
        // (profiles/r6_retok_ab.jsonl); "unigram_workgroup" 2 / 0 force one of them
        if (r->t.kind == ZETT_RETOK_UNIGRAM && (r->unigram_wg == 2 || (r->unigram_wg == 1 && n_tokens <= 32768)))
            hipLaunchKernelGGL(retok_unigram_kernel, dim3((unsigned)((n_tokens + UG_TOKENS - 1) / UG_TOKENS)), dim3(UG_THREADS), 0, st, r->t, r->raw.as<uint8_t>(),
                               r->raw_off.as<int32_t>(), n_tokens, maxlen, pad_id, out, r->scratch.as<int32_t>(), words + 2, words + 1, call);
        else
            hipLaunchKernelGGL(retok_tokens_kernel, dim3((unsigned)((n_tokens + 63) / 64)), dim3(64), 0, st, r->t, r->raw.as<uint8_t>(),
                               r->raw_off.as<int32_t>(), n_tokens, maxlen, pad_id, out, r->scratch.as<int32_t>(), words + 2, words + 1, call);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(r->host_pinned + 4, words, 24, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipEventRecord(r->done, st));
        return 0;
    };
    if (int rc = enqueue()) {
        (void)hipStreamSynchronize(st);
        r->calls--;
        r->recent.pop_back();
        if (r->calls == 0) r->words_ready = false;
        return rc;
    }
    return 0;
}

int zett_retok_result(zett_retok* r, int64_t* n_truncated, int64_t* bad_call, int64_t* bad_token) {
    using namespace zett;
    if (!r || !n_truncated) return fail(ZETT_E_INVALID, "null argument");
    *n_truncated = 0;
    if (bad_call) *bad_call = -1;
    if (bad_token) *bad_token = -1;
    if (r->calls == 0) return 0;
    ZETT_ON_DEVICE(r->device);
    HIP_TRY(hipEventSynchronize(r->done));
    unsigned long long w[3];
    memcpy(w, r->host_pinned + 4, 24);
    const std::vector<zett_retok::Call> recent = r->recent;
    r->words_ready = false;          // the next call starts from fresh words
    r->calls = 0;
    r->recent.clear();
    if (w[0] != ~0ull) {                                       // KeyError: locate the token of the first bad character
        const uint32_t call = (uint32_t)(w[0] >> 32);
        const int32_t pos = (int32_t)(w[0] & 0xffffffffu);
        int64_t lo = -1;
        if (call < recent.size() && recent[call].offsets == nullptr && (w[0] & 0xffffffffu) == 0xfffffffeu) {
            if (bad_call) *bad_call = call;
            return fail(ZETT_E_INVALID, "call %u: the NUL-separated text does not hold n_tokens - 1 separators", call);
        }
        if (call < recent.size() && recent[call].offsets == nullptr) {
            lo = pos;                                          // NUL-separated text: the kernel reported the token index itself
        } else if (call < recent.size()) {
            const auto& cl = recent[call];
            std::vector<int32_t> ho((size_t)cl.n_tokens + 1);
            HIP_TRY(hipMemcpy(ho.data(), cl.offsets, ((size_t)cl.n_tokens + 1) * 4, hipMemcpyDeviceToHost));
            int64_t hi = cl.n_tokens;
            lo = 0;                                            // last token with offset <= pos
            while (lo + 1 < hi) { const int64_t mid = (lo + hi) / 2; if (ho[mid] <= pos) lo = mid; else hi = mid; }
        }
        if (bad_call) *bad_call = call;
        if (bad_token) *bad_token = lo;
        return fail(ZETT_E_KEY, "token %lld holds a character outside the byte-level table (%s %d)", (long long)lo,
                    (call < recent.size() && recent[call].offsets == nullptr) ? "token" : "text offset", pos);
    }
    if (w[1] != ~0ull) {
        const int64_t tok = (int64_t)(w[1] & 0xffffffffu);
        if (bad_call) *bad_call = (int64_t)(w[1] >> 32);
        if (bad_token) *bad_token = tok;
        if (r->t.kind == ZETT_RETOK_WORDPIECE)
            return fail(ZETT_E_STATE, "WordPiece error: Missing [UNK] token from the vocabulary (token %lld)", (long long)tok);
        return fail(ZETT_E_STATE, "Encountered an unknown token but `unk_id` is missing (token %lld)", (long long)tok);
    }
    *n_truncated = (int64_t)w[2];
    return 0;
}

int zett_retokenize(zett_retok* r, const uint8_t* token_chars, const int32_t* offsets, int64_t n_tokens, int32_t maxlen,
                    int32_t pad_id, int32_t* out, int64_t* n_truncated, int64_t* bad_token, void* stream) {
    using namespace zett;
    if (!r || !n_truncated) return fail(ZETT_E_INVALID, "null argument");
    if (n_tokens < 0 || maxlen < 1) return fail(ZETT_E_INVALID, "bad shape");
    if (bad_token) *bad_token = -1;
    *n_truncated = 0;
    if (n_tokens == 0) return 0;
    if (!offsets || !out) return fail(ZETT_E_INVALID, "null argument");
    ZETT_ON_DEVICE(r->device);
    hipStream_t st = (hipStream_t)stream;
    if (r->calls) {                  // results of earlier asynchronous calls nobody asked for: drop them
        int64_t t, c, b;
        (void)zett_retok_result(r, &t, &c, &b);
    }
    // total text length = offsets[n_tokens]
    int32_t* hp = r->host_pinned;
    HIP_TRY(hipMemcpyAsync(hp, offsets + n_tokens, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const int64_t n_text = hp[0];
    if (n_text < 0) return fail(ZETT_E_INVALID, "offsets must be non-decreasing from 0");
    if (int rc = zett_retokenize_async(r, token_chars, offsets, n_tokens, n_text, maxlen, pad_id, out, stream)) return rc;
    return zett_retok_result(r, n_truncated, nullptr, bad_token);
}

}  // extern "C"
