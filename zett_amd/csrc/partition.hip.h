// partition.hip.h — id-affinity row partition for the vocabulary-sharded path (zett_partition_rows, include/zett_hip.h).
//
// Rows of a surface-form matrix are independent, and the reference hands them to its devices in whatever order its random
// permutation left them (scripts/transfer.py:54-67, 90-91; zett/utils.py:26).  Which rows a rank gets is therefore free — and it
// decides how much work the rank's forward is: the hoisted input projection runs once per DISTINCT source id of the shard, and
// layer 0's Q/K/V once per distinct (id, position) pair (levers 2 and 4).  Contiguous shards of the 32 768-row
// headline vocabulary hold 8 340 distinct ids per rank at 8 ranks (29 187 / 8 = 3 648 would be a perfect split) and repeat
// almost no pair.  This kernel assigns rows to ranks so that rows sharing ids share a rank.
//
// Algorithm (deterministic: every rank runs it on the same matrix and must get the same answer — ballots and scans only, no
// order-dependent atomics): ONE workgroup of 1024 threads walks the rows in rounds of 1024.  In a round every thread scores its
// row against each rank — 8 per id of the row the rank already holds (a byte per id: bit r = rank r holds it, L2-resident), 3 for
// the row's "home" rank (first id mod P: rows with the same first id agree on it even within a round, where they cannot see
// each other's choice), minus 0..4 for the rank's fill — and picks the best rank that still has room (ties: the lowest).  Rows are
// admitted per rank in thread order up to the rank's capacity (a wavefront ballot + a 16-entry cross-wave scan per rank); the
// few that find their rank full in that round fill the remaining room in rank order.  Then the round's ids are OR-ed into the
// rank bytes.  Capacities are given by the caller (the row counts its row blocks hand to each rank), so the result drops into
// the existing block exchange.  A final pass writes the row indices grouped by rank, ascending within a rank.
//
// Measured quality (tools/partition_quality.py, 8 ranks): headline vocabulary 8 369 -> 5 971 distinct ids per rank and 0.953 ->
// 0.823 pairs per position (the pair lever's 0.85 threshold is met again); Mistral -> NeoX 11 904 -> 8 083.  A sequential greedy
// on the host reaches 5 644 but costs 3 ms of host time on the critical path; this kernel is ~4 us per round.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace zett {

constexpr int PART_THREADS = 1024;
constexpr int PART_MAX_RANKS = 8;

__global__ __launch_bounds__(PART_THREADS) void partition_rows_kernel(const int32_t* __restrict__ sfm, int64_t n_rows, int seq, int pad,
                                                                     int n_ids, int world, const int32_t* __restrict__ caps,
                                                                     uint32_t* __restrict__ have /* ceil(n_ids / 4) words, zeroed */,
                                                                     int8_t* __restrict__ rank_of /* [n_rows] */,
                                                                     int32_t* __restrict__ perm /* [n_rows] */) {
    __shared__ int s_cnt[PART_MAX_RANKS], s_cap[PART_MAX_RANKS], s_base[PART_MAX_RANKS];
    __shared__ int s_wave[PART_THREADS / 64][PART_MAX_RANKS + 1];      // per wave: choosers of rank r; [.][P]: rows left over
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = world;
    if (tid < PART_MAX_RANKS) { s_cnt[tid] = 0; s_cap[tid] = tid < P ? caps[tid] : 0; }
    __syncthreads();
    if (tid == 0) { int b = 0; for (int r = 0; r < PART_MAX_RANKS; ++r) { s_base[r] = b; b += s_cap[r]; } }
    const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));

    for (int64_t row0 = 0; row0 < n_rows; row0 += PART_THREADS) {
        const int64_t row = row0 + tid;
        const bool valid = row < n_rows;
        int sc[PART_MAX_RANKS];
#pragma unroll
        for (int r = 0; r < PART_MAX_RANKS; ++r) sc[r] = 0;
        int first = 0;
        if (valid) {
            const int32_t* ids = sfm + row * seq;
            for (int j = 0; j < seq; ++j) {
                const int id = ids[j];
                if (id == pad || id < 0 || id >= n_ids) continue;
                if (j == 0) first = id;
                // (agent-scope load: the bytes are written by atomics at L2; a plain load could hit a stale line of this CU's vector cache)
                const uint32_t bits = (__hip_atomic_load(&have[id >> 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> ((id & 3) * 8)) & 0xffu;
#pragma unroll
                for (int r = 0; r < PART_MAX_RANKS; ++r) sc[r] += (bits >> r) & 1;
            }
        }
        // the best rank with room (ties: the lowest)
        int choice = -1, best = -(1 << 30);
        const int home = first % P;
#pragma unroll
        for (int r = 0; r < PART_MAX_RANKS; ++r) {
            if (r >= P) break;
            const int cap = s_cap[r], cnt = s_cnt[r];
            if (cnt >= cap) continue;
            const int s = sc[r] * 8 + (r == home ? 3 : 0) - (cnt * 4) / (cap > 0 ? cap : 1);
            if (s > best) { best = s; choice = r; }
        }
        if (!valid) choice = -1;
        // admission in thread order, per rank
        int my_idx = 0;
        for (int r = 0; r < P; ++r) {
            const unsigned long long m = __ballot(choice == r);
            if (choice == r) my_idx = __popcll(m & lt_mask);
            if (lane == 0) s_wave[wave][r] = __popcll(m);
        }
        __syncthreads();
        bool admitted = false;
        if (choice >= 0) {
            int before = 0;
            for (int w = 0; w < wave; ++w) before += s_wave[w][choice];
            admitted = before + my_idx < s_cap[choice] - s_cnt[choice];
        }
        int final_rank = admitted ? choice : -1;
        // rows left over (their rank filled up within this round): numbered in thread order
        const bool left = valid && !admitted;
        const unsigned long long lm = __ballot(left);
        const int left_idx = __popcll(lm & lt_mask);
        if (lane == 0) s_wave[wave][PART_MAX_RANKS] = __popcll(lm);
        __syncthreads();
        if (tid < P) {          // the round's admissions of rank tid
            int tot = 0;
            for (int w = 0; w < PART_THREADS / 64; ++w) tot += s_wave[w][tid];
            const int room = s_cap[tid] - s_cnt[tid];
            s_cnt[tid] += tot < room ? tot : room;
        }
        __syncthreads();
        if (left) {             // fill the remaining room in rank order
            int j = left_idx;
            for (int w = 0; w < wave; ++w) j += s_wave[w][PART_MAX_RANKS];
            int cum = 0;
            for (int r = 0; r < P; ++r) {
                cum += s_cap[r] - s_cnt[r];
                if (j < cum) { final_rank = r; break; }
            }
        }
        __syncthreads();
        if (tid == 0) {         // ... and count them
            int total_left = 0;
            for (int w = 0; w < PART_THREADS / 64; ++w) total_left += s_wave[w][PART_MAX_RANKS];
            for (int r = 0; r < P && total_left > 0; ++r) {
                const int room = s_cap[r] - s_cnt[r];
                const int take = total_left < room ? total_left : room;
                s_cnt[r] += take;
                total_left -= take;
            }
        }
        if (valid && final_rank >= 0) {
            rank_of[row] = (int8_t)final_rank;
            const int32_t* ids = sfm + row * seq;
            for (int j = 0; j < seq; ++j) {
                const int id = ids[j];
                if (id == pad || id < 0 || id >= n_ids) continue;
                atomicOr(&have[id >> 2], 1u << ((id & 3) * 8 + final_rank));
            }
        } else if (valid) {
            rank_of[row] = (int8_t)(P - 1);          // (unreachable when the capacities sum to n_rows)
        }
        __threadfence();
        __syncthreads();
    }

    // row indices grouped by rank, ascending within a rank
    if (tid < PART_MAX_RANKS) s_cnt[tid] = 0;
    __syncthreads();
    for (int64_t row0 = 0; row0 < n_rows; row0 += PART_THREADS) {
        const int64_t row = row0 + tid;
        const int r_mine = row < n_rows ? (int)rank_of[row] : -1;
        int my_idx = 0;
        for (int r = 0; r < P; ++r) {
            const unsigned long long m = __ballot(r_mine == r);
            if (r_mine == r) my_idx = __popcll(m & lt_mask);
            if (lane == 0) s_wave[wave][r] = __popcll(m);
        }
        __syncthreads();
        if (r_mine >= 0) {
            int before = 0;
            for (int w = 0; w < wave; ++w) before += s_wave[w][r_mine];
            perm[s_base[r_mine] + s_cnt[r_mine] + before + my_idx] = (int32_t)row;
        }
        __syncthreads();
        if (tid < P) {
            int tot = 0;
            for (int w = 0; w < PART_THREADS / 64; ++w) tot += s_wave[w][tid];
            s_cnt[tid] += tot;
        }
        __syncthreads();
    }
}

inline size_t partition_workspace_bytes(int64_t n_rows, int n_ids) {
    return ((size_t)(n_ids + 3) / 4) * 4 + 64 + (((size_t)n_rows + 63) / 64) * 64 + PART_MAX_RANKS * 4;
}

}  // namespace zett
