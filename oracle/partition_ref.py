"""Checker (test infrastructure only: imported by tests/ and tools/, never by zett_amd/) for zett_partition_rows
(zett_amd/csrc/partition.hip.h): a numpy restatement of the SAME batch-synchronous assignment, round by round, so that the
kernel's permutation can be demanded bit for bit.  There is no reference counterpart — the reference hands its devices the
rows of a random permutation in order (scripts/transfer.py:54-67, 90-91; zett/utils.py:26); what is pinned against the
reference is that the row ORDER is free (rows are independent: tests/test_invariants_gpu.py) — parity of this file is
"kernel == its own specification", stated here and in DESIGN.md section 6.
"""
import numpy as np

ROUND = 1024


def partition_rows(ids: np.ndarray, pad: int, n_ids: int, caps) -> np.ndarray:
    """Row indices grouped by rank (rank r's rows at [sum(caps[:r]), +caps[r])), in the order they were assigned: round by round,
    within a round the admitted rows in row order, then the left-over ones."""
    ids = np.asarray(ids)
    n, _ = ids.shape
    caps = np.asarray(caps, dtype=np.int64)
    p = len(caps)
    assert caps.sum() == n
    have = np.zeros((n_ids, p), dtype=bool)
    cnt = np.zeros(p, dtype=np.int64)
    pos = np.zeros(p, dtype=np.int64)                          # packed positions per rank so far
    rank_of = np.full(n, -1, dtype=np.int64)
    groups = [[] for _ in range(p)]
    for s in range(0, n, ROUND):
        rows = ids[s:s + ROUND]
        m = (rows != pad) & (rows >= 0) & (rows < n_ids)
        m63 = m & (np.cumsum(m, axis=1) <= 63)                # (the kernel's byte counters: a row's first 63 usable ids count)
        shared = (have[np.where(m63, rows, 0)] & m63[..., None]).sum(1)
        first = np.where(m[:, 0], rows[:, 0], 0)
        lead = np.clip((pos - pos.sum() // p) // 256, -4, 4)   # a rank's lead in packed positions over the mean (floor division)
        score = shared * 8 + 3 * (np.arange(p)[None, :] == (first % p)[:, None]) - (cnt * 4 // np.maximum(caps, 1) + lead)[None, :]
        score = np.where((cnt < caps)[None, :], score, -(1 << 30))
        choice = np.argmax(score, axis=1)                    # ties: the lowest rank
        assigned = np.full(len(rows), -1, dtype=np.int64)
        for r in range(p):
            sel = np.flatnonzero(choice == r)
            take = sel[:max(int(caps[r] - cnt[r]), 0)]
            assigned[take] = r
            cnt[r] += len(take)
            groups[r].extend((s + take).tolist())
        left = np.flatnonzero(assigned < 0)
        if len(left):
            cum = np.cumsum(caps - cnt)
            rr = np.searchsorted(cum, np.arange(len(left)), side="right")
            assigned[left] = rr
            np.add.at(cnt, rr, 1)
            for i, r in zip(left.tolist(), rr.tolist()):
                groups[r].append(s + i)
        rank_of[s:s + ROUND] = assigned
        np.add.at(pos, assigned, np.minimum(m.sum(1), 63))
        for r in range(p):
            sub = rows[assigned == r]
            mm = (sub != pad) & (sub >= 0) & (sub < n_ids)
            have[sub[mm], r] = True
    return np.concatenate([np.asarray(g, dtype=np.int64) for g in groups]).astype(np.int32)


def shard_statistics(ids: np.ndarray, pad: int, groups):
    """Per group of row indices: (rows, packed positions, distinct ids, distinct (id, position) pairs)."""
    out = []
    for g in groups:
        sub = ids[g]
        m = sub != pad
        pos = np.broadcast_to(np.arange(sub.shape[1]), sub.shape)
        out.append((len(sub), int(m.sum()), len(np.unique(sub[m])), len(np.unique(sub[m].astype(np.int64) * 64 + pos[m]))))
    return out
