// partition.hip.h — id-affinity row partition for the vocabulary-sharded path (zett_partition_rows, include/zett_hip.h).
//
// Rows of a surface-form matrix are independent, and the reference hands them to its devices in whatever order its random
// permutation left them (scripts/transfer.py:54-67, 90-91; zett/utils.py:26).  Which rows a rank gets is therefore free — and it
// decides how much work the rank's forward is: the hoisted input projection runs once per DISTINCT source id of the shard, and
// layer 0's Q/K/V once per distinct (id, position) pair (levers 2 and 4).  Contiguous shards of the 32 768-row
// headline vocabulary hold 8 340 distinct ids per rank at 8 ranks (29 187 / 8 = 3 648 would be a perfect split) and repeat
// almost no pair.  This kernel assigns rows to ranks so that rows sharing ids share a rank.
//
// Algorithm (deterministic: every rank runs it on the same matrix and must get the same answer — ballots, scans and commutative
// integer atomics only): ONE workgroup of 1024 threads walks the rows in rounds of 1024.  In a round every thread scores its row
// against each rank — 8 per id of the row the rank already holds (a byte per id, bit r = rank r holds it: in LDS when the id
// range fits, else in global memory behind agent-scope loads), 3 for the row's "home" rank (first id mod P: rows with the same
// first id agree on it even within a round, where they cannot see each other's choice), minus 0..4 for the rank's fill in rows
// and -4..4 for its lead in PACKED POSITIONS over the mean (a shard's cost is its positions: without this term the ranks end up
// 3.5 % apart, with it ~1 %) — and picks the best rank that still has room (ties: the lowest).  Rows are admitted per rank in
// thread order up to the rank's capacity (a wavefront ballot + a 16-entry cross-wave scan per rank); the few that find their rank
// full in that round fill the remaining room in rank order.  Then the round's ids are OR-ed into the rank bytes.  The next round's
// ids are fetched while this one is decided.  Capacities are given by the caller (the row counts its row blocks hand to each
// rank), so the result drops into the existing block exchange.  A row's place in its rank's group is fixed as it is assigned (per round:
// the admitted rows in thread order, then the left-over ones): the permutation is written in the same pass.
//
// Measured quality (tools/partition_quality.py -> profiles/r5_partition_quality.md, 8 ranks): headline vocabulary 8 370 -> 5 785
// distinct ids per rank and 0.953 -> 0.81 pairs per position (the pair lever's 0.85 threshold is met again), positions within
// 1.4 % of the mean; Mistral -> NeoX 11 904 -> 7 720.  A sequential greedy on the host reaches 5 644 but costs 3 ms of host time
// on the critical path.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace zett {

constexpr int PART_THREADS = 1024;
constexpr int PART_MAX_RANKS = 8;

struct PartCaps { int32_t v[PART_MAX_RANKS]; };          // the ranks' capacities travel by value in the kernel-argument segment (no copy, no host-buffer lifetime)

constexpr int PART_REG_IDS = 8;          // ids of a row held in registers (and prefetched a round ahead); longer rows read the rest from memory

// Barrier of the round loop: everything the rounds exchange lives in LDS, so only the LDS counter is waited for — __syncthreads()
// also waits for vmcnt(0), i.e. for the NEXT round's id prefetch and this round's perm stores (a global round trip, ~2 us, at
// every one of a round's four barriers: the kernel ran 9 us per round with it, against ~2 without).
__device__ __forceinline__ void part_barrier() {
    __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0); vmcnt / expcnt untouched
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

template <bool LDS_HAVE>
__global__ __launch_bounds__(PART_THREADS) void partition_rows_kernel(const int32_t* __restrict__ sfm, int64_t n_rows, int seq, int pad,
                                                                     int n_ids, int world, const PartCaps caps,
                                                                     uint32_t* __restrict__ have_global /* ceil(n_ids / 4) words, zeroed (unused with LDS_HAVE) */,
                                                                     int32_t* __restrict__ perm /* [n_rows] */) {
    extern __shared__ uint32_t s_have[];                               // LDS_HAVE: the rank bytes, ceil(n_ids / 4) words
    __shared__ int s_cnt[PART_MAX_RANKS], s_cap[PART_MAX_RANKS], s_base[PART_MAX_RANKS], s_pos[PART_MAX_RANKS];
    __shared__ int s_wave[PART_THREADS / 64][PART_MAX_RANKS + 1];      // per wave: choosers of rank r; [.][P]: rows left over
    __shared__ int s_wpos[PART_THREADS / 64][PART_MAX_RANKS];          // per wave: packed positions of the rows it gave rank r this round
    __shared__ int s_pen[PART_MAX_RANKS];                              // the rank's penalty this round (fill + lead in positions); a full rank: 1 << 28
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = world;
    if (tid < PART_MAX_RANKS) { s_cnt[tid] = 0; s_pos[tid] = 0; s_cap[tid] = tid < P ? caps.v[tid] : 0; }
    if (LDS_HAVE)
        for (int i = tid; i < (n_ids + 3) / 4; i += PART_THREADS) s_have[i] = 0u;
    __syncthreads();
    if (tid == 0) { int b = 0; for (int r = 0; r < PART_MAX_RANKS; ++r) { s_base[r] = b; b += s_cap[r]; } }
    const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    // The per-rank part of the score, once per round by ONE thread instead of by all 1024 (two integer divisions per rank: the
    // kernel is a single workgroup, i.e. bound by the instructions its 16 waves issue on one CU — 9 us per round before this).
    auto penalties = [&]() {
        if (tid == 0) {
            int pos_total = 0;
            for (int r = 0; r < P; ++r) pos_total += s_pos[r];
            const int pos_mean = pos_total / P;
            for (int r = 0; r < PART_MAX_RANKS; ++r) {
                if (r >= P || s_cnt[r] >= s_cap[r]) { s_pen[r] = 1 << 28; continue; }
                const int d = s_pos[r] - pos_mean;
                int lead = d >= 0 ? d / 256 : -((-d + 255) / 256);          // floor division
                lead = lead < -4 ? -4 : (lead > 4 ? 4 : lead);
                s_pen[r] = (s_cnt[r] * 4) / s_cap[r] + lead;
            }
        }
    };
    __syncthreads();
    penalties();
    __syncthreads();
    const int nreg = seq < PART_REG_IDS ? seq : PART_REG_IDS;
    auto usable = [&](int id) { return id != pad && id >= 0 && id < n_ids; };
    auto load_bits = [&](int id) -> uint32_t {
        // (global copy: agent-scope load — the bytes are written by atomics at L2; a plain load could hit a stale line of this CU's vector cache)
        const uint32_t w = LDS_HAVE ? s_have[id >> 2] : __hip_atomic_load(&have_global[id >> 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return (w >> ((id & 3) * 8)) & 0xffu;
    };
    auto mark = [&](int id, int r) {
        if (LDS_HAVE) atomicOr(&s_have[id >> 2], 1u << ((id & 3) * 8 + r));
        else atomicOr(&have_global[id >> 2], 1u << ((id & 3) * 8 + r));
    };
    int next_ids[PART_REG_IDS];
#pragma unroll
    for (int j = 0; j < PART_REG_IDS; ++j) next_ids[j] = (j < nreg && tid < n_rows) ? sfm[(int64_t)tid * seq + j] : pad;

    for (int64_t row0 = 0; row0 < n_rows; row0 += PART_THREADS) {
        const int64_t row = row0 + tid;
        const bool valid = row < n_rows;
        int ids[PART_REG_IDS];
#pragma unroll
        for (int j = 0; j < PART_REG_IDS; ++j) ids[j] = next_ids[j];
        {   // the next round's ids: requested now, consumed a round later
            const int64_t nrow = row + PART_THREADS;
#pragma unroll
            for (int j = 0; j < PART_REG_IDS; ++j) next_ids[j] = (j < nreg && nrow < n_rows) ? sfm[nrow * seq + j] : pad;
        }
        // ids of the row a rank already holds, per rank: eight byte counters in two words (bit r of a rank byte -> byte r & 3 of
        // word r >> 2: x * 0x00204081 & 0x01010101 spreads four bits over four bytes)
        uint32_t sc_lo = 0, sc_hi = 0;
        int npos = 0;                                        // usable positions of the row (counted up to 63 for the balance term)
        const int first = (valid && usable(ids[0])) ? ids[0] : 0;
        auto tally = [&](uint32_t bits) {
            sc_lo += ((bits & 15u) * 0x00204081u) & 0x01010101u;
            sc_hi += ((bits >> 4) * 0x00204081u) & 0x01010101u;
        };
        if (valid) {
#pragma unroll
            for (int j = 0; j < PART_REG_IDS; ++j) {
                if (j < nreg && usable(ids[j])) { tally(load_bits(ids[j])); ++npos; }
            }
            for (int j = PART_REG_IDS; j < seq; ++j) {
                const int id = sfm[row * seq + j];
                if (!usable(id)) continue;
                if (npos < 63) { tally(load_bits(id)); ++npos; }          // (byte counters: a row's first 63 usable ids count)
            }
        }
        // the best rank with room (ties: the lowest)
        int choice = -1, best = -(1 << 30);
        const int home = first % P;
#pragma unroll
        for (int r = 0; r < PART_MAX_RANKS; ++r) {
            const int pen = s_pen[r];
            const int shared = (int)(((r < 4 ? sc_lo : sc_hi) >> ((r & 3) * 8)) & 0xffu);
            const int sv = shared * 8 + (r == home ? 3 : 0) - pen;
            if (pen < (1 << 28) && sv > best) { best = sv; choice = r; }
        }
        if (!valid) choice = -1;
        // admission in thread order, per rank
        int my_idx = 0;
        for (int r = 0; r < P; ++r) {
            const unsigned long long m = __ballot(choice == r);
            if (choice == r) my_idx = __popcll(m & lt_mask);
            if (lane == 0) s_wave[wave][r] = __popcll(m);
        }
        part_barrier();
        bool admitted = false;
        int slot = -1;                                       // the row's place within its rank's group
        if (choice >= 0) {
            int before = 0;
            for (int w = 0; w < wave; ++w) before += s_wave[w][choice];
            admitted = before + my_idx < s_cap[choice] - s_cnt[choice];
            slot = s_cnt[choice] + before + my_idx;
        }
        int final_rank = admitted ? choice : -1;
        // rows left over (their rank filled up within this round): numbered in thread order
        const bool left = valid && !admitted;
        const unsigned long long lm = __ballot(left);
        const int left_idx = __popcll(lm & lt_mask);
        if (lane == 0) s_wave[wave][PART_MAX_RANKS] = __popcll(lm);
        part_barrier();
        int total_left = 0;
        for (int w = 0; w < PART_THREADS / 64; ++w) total_left += s_wave[w][PART_MAX_RANKS];
        if (tid < P) {          // the round's admissions of rank tid
            int tot = 0;
            for (int w = 0; w < PART_THREADS / 64; ++w) tot += s_wave[w][tid];
            const int room = s_cap[tid] - s_cnt[tid];
            s_cnt[tid] += tot < room ? tot : room;
        }
        if (total_left > 0) {   // (uniform) fill the remaining room in rank order, behind the rows admitted above
            part_barrier();
            if (left) {
                int j = left_idx;
                for (int w = 0; w < wave; ++w) j += s_wave[w][PART_MAX_RANKS];
                int cum = 0;
                for (int r = 0; r < P; ++r) {
                    const int room = s_cap[r] - s_cnt[r];
                    if (j < cum + room) { final_rank = r; slot = s_cnt[r] + (j - cum); break; }
                    cum += room;
                }
            }
            part_barrier();
            if (tid == 0) {     // ... and count them
                int tl = total_left;
                for (int r = 0; r < P && tl > 0; ++r) {
                    const int room = s_cap[r] - s_cnt[r];
                    const int take = tl < room ? tl : room;
                    s_cnt[r] += take;
                    tl -= take;
                }
            }
        }
        if (valid && final_rank < 0) { final_rank = P - 1; slot = 0; }          // (unreachable when the capacities sum to n_rows)
        // packed positions the round gave each rank: bit-sliced ballots (positions < 64 per row), a sum per wave, 16 waves added below
        {
            unsigned long long bitm[6];
#pragma unroll
            for (int b = 0; b < 6; ++b) bitm[b] = __ballot(valid && ((npos >> b) & 1));
            for (int r = 0; r < P; ++r) {
                const unsigned long long m = __ballot(valid && final_rank == r);
                int sum = 0;
#pragma unroll
                for (int b = 0; b < 6; ++b) sum += __popcll(bitm[b] & m) << b;
                if (lane == 0) s_wpos[wave][r] = sum;
            }
        }
        if (valid) {
            perm[s_base[final_rank] + slot] = (int32_t)row;
#pragma unroll
            for (int j = 0; j < PART_REG_IDS; ++j)
                if (j < nreg && usable(ids[j])) mark(ids[j], final_rank);
            for (int j = PART_REG_IDS; j < seq; ++j) {
                const int id = sfm[row * seq + j];
                if (usable(id)) mark(id, final_rank);
            }
        }
        if (!LDS_HAVE) { __threadfence(); __syncthreads(); }
        else part_barrier();
        if (tid < P) {
            int tot = 0;
            for (int w = 0; w < PART_THREADS / 64; ++w) tot += s_wpos[w][tid];
            s_pos[tid] += tot;
        }
        part_barrier();
        penalties();
        part_barrier();
    }
}

// dst[order[i]] = src[i] for rows of row_bytes bytes (a multiple of 16, or exactly 4): puts an exchanged block of predicted rows
// back into vocabulary order.  One workgroup per row, 16-byte accesses: an HBM stream.
__global__ __launch_bounds__(256) void scatter_rows_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                           const int64_t* __restrict__ order, int64_t n_rows, int64_t row_bytes) {
    for (int64_t i = blockIdx.x; i < n_rows; i += gridDim.x) {
        const int64_t to = order[i];
        if (to < 0) continue;
        const uint4* s = (const uint4*)(src + i * row_bytes);
        uint4* d = (uint4*)(dst + to * row_bytes);
        for (int64_t c = threadIdx.x; c < row_bytes / 16; c += blockDim.x) d[c] = s[c];
    }
}

__global__ void scatter_words_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, const int64_t* __restrict__ order, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && order[i] >= 0) dst[order[i]] = src[i];
}

inline size_t partition_workspace_bytes(int64_t n_rows, int n_ids) {
    return ((size_t)(n_ids + 3) / 4) * 4 + 64 + (((size_t)n_rows + 63) / 64) * 64 + PART_MAX_RANKS * 4;
}

}  // namespace zett
