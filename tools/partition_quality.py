#!/usr/bin/env python
"""CPU: what the id-affinity row partition (zett_partition_rows; specification: oracle/partition_ref.py) does to a rank's shard on
the BASELINE workloads — distinct source ids (rows of the hoisted input projection), distinct (id, position) pairs per packed
position (the pair lever is taken below 0.85) and packed positions per rank, against contiguous shards.

    python tools/partition_quality.py > profiles/r6_partition_quality.md

r6: beside the greedy of zett_partition_rows, the trivially PARALLEL orders — a stable sort of the rows by a key
of their ids, cut into rank ranges (a device radix sort: ~30-50 us, no sequential rounds): by (first id, second id), and by (the
row's most frequent id over the vocabulary, its second most frequent).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import partition_ref  # noqa: E402
from zett_amd import synth  # noqa: E402

print("| workload | ranks | shards | distinct ids per rank (mean / max) | perfect split | pairs per position (mean) | packed positions per rank (min .. max) |")
print("|---|---|---|---|---|---|---|")
for name in ("mistral_gpt2_32k", "mistral_neox", "tinyllama_neox", "xlmr_gpt2", "llama3_256k"):
    cfg, rows, _, hist = synth.workload(name)
    ids = synth.make_surface_forms(cfg, rows, seed=0, hist=hist)
    pad, n_ids = cfg["pad_token_id"], cfg["original_vocab_size"] + cfg["hn_n_extra_tokens"]
    total = len(np.unique(ids[ids != pad]))
    for world in (2, 4, 8):
        per = -(-rows // world)
        caps = [max(0, min(per, rows - r * per)) for r in range(world)]
        off = np.concatenate([[0], np.cumsum(caps)])
        perm = partition_ref.partition_rows(ids, pad, n_ids, caps)
        key1 = ids[:, 0].astype(np.int64) * (n_ids + 1) + (ids[:, 1].astype(np.int64) if ids.shape[1] > 1 else 0)
        by_first = np.argsort(key1, kind="stable")
        cnt = np.bincount(ids[ids != pad], minlength=n_ids + 1)
        pop = np.where(ids == pad, -1, cnt[np.minimum(ids, n_ids)])
        a1 = pop.argmax(1)
        pop2 = pop.copy(); pop2[np.arange(rows), a1] = -2
        a2 = pop2.argmax(1)
        key2 = ids[np.arange(rows), a1].astype(np.int64) * (n_ids + 1) + ids[np.arange(rows), a2]
        by_pop = np.argsort(key2, kind="stable")
        for label, groups in (("contiguous", [np.arange(off[r], off[r + 1]) for r in range(world)]), ("affinity (greedy rounds, zett_partition_rows)", [perm[off[r]:off[r + 1]] for r in range(world)]),
                              ("sorted by (first id, second id)", [by_first[off[r]:off[r + 1]] for r in range(world)]),
                              ("sorted by (most frequent id, second most frequent)", [by_pop[off[r]:off[r + 1]] for r in range(world)])):
            st = partition_ref.shard_statistics(ids, pad, groups)
            print(f"| {name} ({rows}) | {world} | {label} | {np.mean([s[2] for s in st]):.0f} / {max(s[2] for s in st)} | {total // world} | "
                  f"{np.mean([s[3] / s[1] for s in st]):.3f} | {min(s[1] for s in st)} .. {max(s[1] for s in st)} |")
